// rsx_sraw.hip -- Canon sRaw chroma interpolation + YCbCr -> RGB for gfx950.
//
// Replaces Cr2sRawInterpolator::interpolate (interpolators/Cr2sRawInterpolator.cpp
// :510-542): interpolate_422<v> (:95-186) walks rows of [Y1 Y2 Cb Cr] groups,
// interpolate_420<v> (:188-460) pairs of rows of [Y1 Y2 Y3 Y4 Cb Cr] groups; every
// group ("MCU") yields 2 / 2x2 RGB pixels.  Chroma of the pixels that carry none is
// the mean (>> 1, >> 2, no rounding) of the neighbouring groups' chroma after
// "- 16384 + hue" (:69-83); the last group of a row / the last row copy instead.
// YUV_TO_RGB<v> (:470-506) and STORE_RGB (:462-468) are reproduced in 32-bit
// integer arithmetic.  One lane per 4 groups: 32 or 48 input bytes as 16-byte loads
// (+ one neighbour dword), 48 output bytes per output row as 16-byte stores; pure
// streaming.
#include "rsx_device.h"

namespace rsx {

namespace {

constexpr int SRAW_THREADS = 256;

struct Chroma {
  int cb, cr;
};

// LoadCbCr + signExtend + applyHue (:48-80); `cbcr` = the group's last dword
__device__ __forceinline__ Chroma chroma_of(uint32_t cbcr, int hue) {
  Chroma c;
  c.cb = int(cbcr & 0xFFFFu) - 16384 + hue;
  c.cr = int(cbcr >> 16) - 16384 + hue;
  return c;
}

__device__ __forceinline__ uint32_t clamp16(int v) { // clampBits(v, 16)
  return uint32_t(v < 0 ? 0 : (v > 65535 ? 65535 : v));
}

// YUV_TO_RGB<version> + STORE_RGB; the products wrap like the reference's ints
struct Rgb {
  uint32_t r, g, b;
};
template <int VERSION>
__device__ __forceinline__ Rgb yuv_to_rgb(int Y, Chroma c, const int* coeffs) {
  int r, g, b;
  if (VERSION == 0) { // "Algorithm found in EOS 40D" :472-480
    r = Y + c.cr - 512;
    g = Y + ((-778 * c.cb - (c.cr * 2048)) >> 12) - 512;
    b = Y + (c.cb - 512);
  } else if (VERSION == 1) { // :482-490
    r = Y + ((50 * c.cb + 22929 * c.cr) >> 12);
    g = Y + ((-5640 * c.cb - 11751 * c.cr) >> 12);
    b = Y + ((29040 * c.cb - 101 * c.cr) >> 12);
  } else { // "Algorithm found in EOS 5d Mk III" :492-501
    r = Y + c.cr;
    g = Y + ((-778 * c.cb - (c.cr * 2048)) >> 12);
    b = Y + c.cb;
  }
  r = int(uint32_t(coeffs[0]) * uint32_t(r));
  g = int(uint32_t(coeffs[1]) * uint32_t(g));
  b = int(uint32_t(coeffs[2]) * uint32_t(b));
  return {clamp16(r >> 8), clamp16(g >> 8), clamp16(b >> 8)};
}

__device__ __forceinline__ Chroma mean2(Chroma a, Chroma b) { // :82-87
  return {(a.cb + b.cb) >> 1, (a.cr + b.cr) >> 1};
}
__device__ __forceinline__ Chroma mean4(Chroma a, Chroma b, Chroma c, Chroma d) { // :89-95
  return {(a.cb + b.cb + c.cb + d.cb) >> 2, (a.cr + b.cr + c.cr + d.cr) >> 2};
}

// Chroma of the groups m0 .. m0+4 of one input row (the fifth is the right
// neighbour of the fourth; past the end of the row it repeats the last group, which
// makes "mean with the neighbour" the copy the reference does there).
template <int GW>
__device__ __forceinline__ void load_row(const uint32_t* __restrict__ I, uint32_t m0,
                                         uint32_t num_mcus, int hue, uint32_t (&w)[4 * GW],
                                         Chroma (&c)[5]) {
  if (m0 + 4 <= num_mcus) {
    // 4 groups = GW 16-byte pieces (rows start 16-byte aligned)
    const uint4* __restrict__ v = reinterpret_cast<const uint4*>(I + GW * m0);
#pragma unroll
    for (int k = 0; k < GW; ++k) {
      const uint4 t = v[k];
      w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4 * GW; ++k)
      w[k] = (m0 + k / GW) < num_mcus ? I[GW * m0 + k] : 0u;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    c[k] = chroma_of(w[GW * k + GW - 1], hue);
  const uint32_t last = num_mcus - 1;
  const uint32_t m4 = m0 + 4 <= last ? m0 + 4 : last;
  c[4] = chroma_of(I[GW * m4 + GW - 1], hue);
#pragma unroll
  for (int k = 0; k < 4; ++k) // groups past the end of the row: harmless copies
    if (m0 + k > last)
      c[k] = c[4];
}

// 12 dwords (4 groups x 2 pixels x 3 samples) of one output row
__device__ __forceinline__ void store_row(uint32_t* __restrict__ O, uint32_t m0,
                                          uint32_t num_mcus, const uint32_t (&o)[12]) {
  if (m0 + 4 <= num_mcus) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* __restrict__ v = reinterpret_cast<u32x4*>(O + 3 * m0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      u32x4 t;
      t.x = o[4 * k]; t.y = o[4 * k + 1]; t.z = o[4 * k + 2]; t.w = o[4 * k + 3];
      __builtin_nontemporal_store(t, v + k); // written once, never re-read here
    }
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (m0 + k / 3 < num_mcus)
        O[3 * m0 + k] = o[k];
  }
}

__device__ __forceinline__ void pack2(uint32_t* o, Rgb p, Rgb q) {
  o[0] = p.r | (p.g << 16);
  o[1] = p.b | (q.r << 16);
  o[2] = q.g | (q.b << 16);
}

// One lane = 4 consecutive groups of one input row: 32 / 48 input bytes as 16-byte
// loads, 48 output bytes per output row as 16-byte stores.
template <int VERSION>
__global__ __launch_bounds__(SRAW_THREADS) void sraw_kernel(
    const SrawJobDev* __restrict__ jobs, const uint32_t* __restrict__ job_block_start,
    int n_jobs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base) {
  // block -> job
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (job_block_start[mid] <= blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const SrawJobDev J = jobs[lo];
  if (int(J.version) != VERSION)
    return;
  const uint32_t local = blockIdx.x - job_block_start[lo];
  const uint32_t r = local / J.blocks_per_row;
  const uint32_t m0 = 4 * ((local - r * J.blocks_per_row) * SRAW_THREADS + threadIdx.x);
  if (m0 >= J.num_mcus)
    return;
  const int coeffs[3] = {J.coeffs[0], J.coeffs[1], J.coeffs[2]};
  const uint32_t* __restrict__ I0 =
      reinterpret_cast<const uint32_t*>(in_base + J.in_offset + uint64_t(r) * J.in_pitch);
  uint32_t o[12];
  if (J.gs == 4) {
    // 4:2:2 -- interpolate_422_row :95-176: [Y1|Y2, Cb|Cr] per group
    uint32_t w[8];
    Chroma c[5];
    load_row<2>(I0, m0, J.num_mcus, J.hue, w, c);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      pack2(o + 3 * k, yuv_to_rgb<VERSION>(int(w[2 * k] & 0xFFFFu), c[k], coeffs),
            yuv_to_rgb<VERSION>(int(w[2 * k] >> 16), mean2(c[k], c[k + 1]), coeffs));
    store_row(reinterpret_cast<uint32_t*>(out_base + J.out_offset + uint64_t(r) * J.out_pitch),
              m0, J.num_mcus, o);
    return;
  }
  // 4:2:0 -- interpolate_420_row :188-345 and the last two lines :385-460:
  // [Y1|Y2, Y3|Y4, Cb|Cr] per group; the row below supplies the vertical means
  // (the last row is its own "row below": mean(x, x) == x is the copy :418-420)
  uint32_t w[12], wd[12];
  Chroma c[5], d[5];
  load_row<3>(I0, m0, J.num_mcus, J.hue, w, c);
  const uint32_t rd = r + 1 < J.rows ? r + 1 : r;
  load_row<3>(reinterpret_cast<const uint32_t*>(in_base + J.in_offset + uint64_t(rd) * J.in_pitch),
              m0, J.num_mcus, J.hue, wd, d);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    pack2(o + 3 * k, yuv_to_rgb<VERSION>(int(w[3 * k] & 0xFFFFu), c[k], coeffs),
          yuv_to_rgb<VERSION>(int(w[3 * k] >> 16), mean2(c[k], c[k + 1]), coeffs));
  store_row(reinterpret_cast<uint32_t*>(out_base + J.out_offset + uint64_t(2 * r) * J.out_pitch),
            m0, J.num_mcus, o);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    pack2(o + 3 * k,
          yuv_to_rgb<VERSION>(int(w[3 * k + 1] & 0xFFFFu), mean2(c[k], d[k]), coeffs),
          yuv_to_rgb<VERSION>(int(w[3 * k + 1] >> 16),
                              mean4(c[k], c[k + 1], d[k], d[k + 1]), coeffs));
  store_row(
      reinterpret_cast<uint32_t*>(out_base + J.out_offset + uint64_t(2 * r + 1) * J.out_pitch),
      m0, J.num_mcus, o);
}

} // namespace

uint32_t sraw_blocks_for(SrawJobDev* j) {
  j->blocks_per_row = (j->num_mcus + 4 * SRAW_THREADS - 1) / (4 * SRAW_THREADS);
  return j->rows * j->blocks_per_row;
}

hipError_t launch_sraw(const SrawJobDev* d_jobs, const uint32_t* d_block_start, int n_jobs,
                       uint32_t total_blocks, const bool versions[3], const void* in_base,
                       void* out_base, hipStream_t stream) {
  if (total_blocks == 0)
    return hipSuccess;
  const dim3 grid(total_blocks), block(SRAW_THREADS);
  const uint8_t* in = static_cast<const uint8_t*>(in_base);
  uint8_t* out = static_cast<uint8_t*>(out_base);
  if (versions[0])
    hipLaunchKernelGGL(sraw_kernel<0>, grid, block, 0, stream, d_jobs, d_block_start, n_jobs,
                       in, out);
  if (versions[1])
    hipLaunchKernelGGL(sraw_kernel<1>, grid, block, 0, stream, d_jobs, d_block_start, n_jobs,
                       in, out);
  if (versions[2])
    hipLaunchKernelGGL(sraw_kernel<2>, grid, block, 0, stream, d_jobs, d_block_start, n_jobs,
                       in, out);
  return hipGetLastError();
}

} // namespace rsx
