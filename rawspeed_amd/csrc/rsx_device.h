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
  uint32_t seg_groups;   // groups per segment (rows are split evenly)
  uint32_t out_aligned;  // every output row start is 16-byte aligned
  uint32_t post_shift;   // UNPACK_MODE_SHIFT: samples are shifted right by this;
                         // UNPACK_MODE_LUT8: index of the job's table
};

// Launch flavours.  PACKED is decodePackedInt; SHIFT is the same stream walk
// followed by `>> post_shift` (decode12BitRawUnpackedLeftAligned); CONTROL is
// decode12BitRawWithControl (bps field = 1 for big-endian nibble order).
// FP: F32 images -- bps 16 / 24 widened to binary32 (decodePackedFP), bps 32 copied.
enum UnpackMode {
  UNPACK_MODE_PACKED = 0,
  UNPACK_MODE_SHIFT = 1,
  UNPACK_MODE_CONTROL = 2,
  UNPACK_MODE_FP = 3,
  UNPACK_MODE_LUT8 = 4 // decode8BitRaw<false>: 8-bit walk + 256-entry table
};

size_t unpack_lds_bytes();
// fill groups_per_row / segs_per_row / seg_groups from n_rows and cols;
// return the number of workgroups of the job
uint32_t unpack_blocks_for(UnpackJobDev* u);
const char* unpack_kernel_name();
uint32_t unpack_control_blocks_for(UnpackJobDev* u);
uint32_t unpack_fp_blocks_for(UnpackJobDev* u);
hipError_t launch_unpack_mode(int mode, int order, const UnpackJobDev* d_jobs,
                              const uint32_t* d_block_start, int n_jobs,
                              uint32_t total_blocks, const void* in_base,
                              void* out_base, hipStream_t stream);
hipError_t launch_unpack(int order, const UnpackJobDev* d_jobs,
                         const uint32_t* d_block_start, int n_jobs,
                         uint32_t total_blocks, const void* in_base,
                         void* out_base, hipStream_t stream);

// One Cr2sRawInterpolator job, flattened for the kernel (rsx_sraw.hip).
struct SrawJobDev {
  uint64_t in_offset;  // subsampled image, relative to the plan's input base
  uint64_t out_offset; // interpolated image, relative to the output base
  uint32_t in_pitch, out_pitch; // bytes
  uint32_t rows;       // input rows
  uint32_t num_mcus;   // groups per input row
  uint32_t gs;         // samples per group: 4 (4:2:2) or 6 (4:2:0)
  uint32_t version;
  int32_t coeffs[3];
  int32_t hue;
  uint32_t blocks_per_row;
};

// measurement aid: plain streaming kernel moving in_bytes in and out_bytes out
hipError_t launch_stream_probe(const void* in, uint64_t in_bytes, void* out,
                               uint64_t out_bytes, hipStream_t stream);

uint32_t sraw_blocks_for(SrawJobDev* j);
hipError_t launch_sraw(const SrawJobDev* d_jobs, const uint32_t* d_block_start, int n_jobs,
                       uint32_t total_blocks, const bool versions[3], const void* in_base,
                       void* out_base, hipStream_t stream);

} // namespace rsx
