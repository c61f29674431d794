// rsx_device.h -- structures shared between the host launchers and the HIP
// kernels, plus the launcher prototypes.
#pragma once

#include "rsx.h"

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace rsx {

// One strip of UncompressedDecompressor work, flattened for the kernel.
struct UnpackJobDev {
  uint64_t in_offset;    // strip start, relative to the batch input base
  uint64_t stream_bytes; // crop_h * pitch: bytes the bit stream may read
  uint64_t out_offset;   // first output row (and column), relative to output base
  uint32_t in_pitch;
  uint32_t out_pitch;    // bytes
  uint32_t n_rows;       // rows decoded = min(crop_h, dim_y - crop_y)
  uint32_t cols;         // samples per row = crop_w * cpp
  uint32_t bps;
  uint32_t groups_per_row; // ceil(cols / 8)
  uint32_t segs_per_row;
  uint32_t out_aligned;  // every output row start is 16-byte aligned
};

size_t unpack_lds_bytes();
uint32_t unpack_blocks_for(uint32_t n_rows, uint32_t cols, uint32_t* segs_per_row,
                           uint32_t* groups_per_row);
const char* unpack_kernel_name();
hipError_t launch_unpack(int order, const UnpackJobDev* d_jobs,
                         const uint32_t* d_block_start, int n_jobs,
                         uint32_t total_blocks, const void* in_base,
                         void* out_base, hipStream_t stream);

} // namespace rsx
