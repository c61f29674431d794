// rsx_unpack.hip -- packed-integer unpack kernel for gfx950 (MI355X).
//
// Replaces the scalar loop of UncompressedDecompressor::decodePackedInt<Pump>
// (decompressors/UncompressedDecompressor.cpp:188-200): one bit stream over the
// whole strip, out(row, x) = getBits(bps), skipBytes(pitch - rowbytes) per row.
//
// Closed form used here (verified against the reference): the
// sample (r, x) is the `bps` bits at stream bit position 8*r*pitch + x*bps in
// consumption order; the stream byte s maps to memory byte
//   LSB / MSB : s            (BitStreamLSB.h / BitStreamMSB.h :31-43)
//   MSB16     : s ^ 1        (u16 little-endian chunks, BitStreamMSB16.h)
//   MSB32     : s ^ 3        (u32 little-endian chunks, BitStreamMSB32.h)
// and bytes past the end of the strip read as zero (BitStreamer.h:100-132).
//
// HBM-streaming design: a workgroup of 256 lanes owns one segment of up to
// 1024 eight-sample groups of one row.  The packed bytes of the segment are
// fetched with 16-byte coalesced loads (all issued before the first use),
// staged through LDS, and every lane then re-reads the <= 20 bytes of its
// groups from LDS, extracts 8 samples with 64-bit shifts and emits one
// 16-byte coalesced store per group.  No MFMA: there is no contraction here.
#include "rsx_device.h"

namespace rsx {

namespace {

// build-time tuning knobs (A/B measured on MI355X)
#ifndef RSX_UNPACK_GPT
#define RSX_UNPACK_GPT 4
#endif
#ifndef RSX_UNPACK_NT
#define RSX_UNPACK_NT 3 // bit 0: non-temporal loads, bit 1: non-temporal stores (measured: +6..10 %)
#endif
constexpr int UNPACK_THREADS = 256;
constexpr int GROUPS_PER_THREAD = RSX_UNPACK_GPT;
constexpr int SEG_GROUPS = UNPACK_THREADS * GROUPS_PER_THREAD; // 1024 groups = 8192 samples
// worst case: 1024 groups * 16 bits = 16384 B, + 15 B misalignment of the
// segment start + 20 B over-read of the last lane, rounded to 16 B chunks
constexpr int SEG_CHUNKS = (SEG_GROUPS * 16 + 16 + 32) / 16; // 1027
constexpr int CHUNKS_PER_THREAD = (SEG_CHUNKS + UNPACK_THREADS - 1) / UNPACK_THREADS; // 5

// 16 bytes at strip offset `off` with zero fill outside [0, stream_bytes).
// `aligned16`: the strip base is 16-byte aligned, so one dwordx4 load does it.
__device__ __forceinline__ uint4 load_chunk(const uint8_t* __restrict__ base,
                                            int64_t off, int64_t stream_bytes,
                                            bool aligned16) {
  if (off >= 0 && off + 16 <= stream_bytes && aligned16) {
#if RSX_UNPACK_NT & 1
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off));
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<const uint4*>(base + off);
#endif
  }
  uint32_t w[4] = {0, 0, 0, 0};
  if (off < stream_bytes && off + 16 > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t o = off + i;
      const uint32_t b = (o >= 0 && o < stream_bytes) ? base[o] : 0u;
      w[i >> 2] |= b << (8 * (i & 3));
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// ORDER: 0 LSB, 1 MSB, 2 MSB16, 3 MSB32
template <int ORDER>
__global__ __launch_bounds__(UNPACK_THREADS) void unpack_kernel(
    const UnpackJobDev* __restrict__ jobs, const uint32_t* __restrict__ job_block_start,
    int n_jobs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);

  // block -> job (jobs of one launch share ORDER; block ranges are prefix sums)
  int job = 0;
  {
    int lo = 0, hi = n_jobs - 1;
    const uint32_t b = blockIdx.x;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (job_block_start[mid] <= b)
        lo = mid;
      else
        hi = mid - 1;
    }
    job = lo;
  }
  const UnpackJobDev J = jobs[job];
  const uint32_t local_block = blockIdx.x - job_block_start[job];
  const uint32_t row = local_block / J.segs_per_row;
  const uint32_t seg = local_block - row * J.segs_per_row;

  const uint32_t bps = J.bps;
  const uint32_t g0 = seg * SEG_GROUPS; // first group of this segment
  uint32_t seg_groups = J.groups_per_row - g0;
  if (seg_groups > SEG_GROUPS)
    seg_groups = SEG_GROUPS;

  const uint8_t* __restrict__ in = in_base + J.in_offset;
  const int64_t stream_bytes = J.stream_bytes;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(in) & 15) == 0;

  // strip-relative byte range of this segment
  const int64_t start = int64_t(row) * J.in_pitch + int64_t(g0) * bps;
  const int64_t a0 = start & ~int64_t(15);
  const uint32_t lead = uint32_t(start - a0);
  // bytes needed: all groups + 20 bytes over-read window of the last lane
  const uint32_t need = lead + seg_groups * bps + 20;
  const uint32_t n_chunks = (need + 15) >> 4;

  // ---- stage: all global loads first, then the LDS writes ----------------
  uint4 v[CHUNKS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CHUNKS_PER_THREAD; ++k) {
    const uint32_t c = threadIdx.x + k * UNPACK_THREADS;
    if (c < n_chunks)
      v[k] = load_chunk(in, a0 + int64_t(c) * 16, stream_bytes, aligned16);
  }
#pragma unroll
  for (int k = 0; k < CHUNKS_PER_THREAD; ++k) {
    const uint32_t c = threadIdx.x + k * UNPACK_THREADS;
    if (c < n_chunks)
      *reinterpret_cast<uint4*>(smem + c * 16) = v[k];
  }
  __syncthreads();

  // ---- extract -----------------------------------------------------------
  uint8_t* __restrict__ out_row = out_base + J.out_offset + uint64_t(row) * J.out_pitch;
  const bool out_aligned = J.out_aligned != 0;
  const uint32_t mask = (1u << bps) - 1u;

#pragma unroll
  for (int k = 0; k < GROUPS_PER_THREAD; ++k) {
    const uint32_t gl = threadIdx.x + k * UNPACK_THREADS; // group within segment
    if (gl >= seg_groups)
      break;
    const uint32_t ob = lead + gl * bps; // LDS byte offset of the group
    const uint32_t wi = ob >> 2;
    const uint32_t kb = ob & 3; // byte shift inside the first dword
    uint32_t d0 = lds[wi], d1 = lds[wi + 1], d2 = lds[wi + 2], d3 = lds[wi + 3],
             d4 = lds[wi + 4];
    uint32_t s[8];
    if (ORDER == 0) {
      // little-endian bit stream: drop `kb` low bytes, then peel from the bottom
      const uint32_t sel = 0x03020100u + 0x01010101u * kb;
      const uint32_t n0 = __builtin_amdgcn_perm(d1, d0, sel);
      const uint32_t n1 = __builtin_amdgcn_perm(d2, d1, sel);
      const uint32_t n2 = __builtin_amdgcn_perm(d3, d2, sel);
      const uint32_t n3 = __builtin_amdgcn_perm(d4, d3, sel);
      uint64_t lo = (uint64_t(n1) << 32) | n0;
      uint64_t hi = (uint64_t(n3) << 32) | n2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i] = uint32_t(lo) & mask;
        lo = (lo >> bps) | (hi << (64 - bps));
        hi >>= bps;
      }
    } else {
      // big-endian view of the stream dwords for this chunk order
      if (ORDER == 1) {
        d0 = __builtin_bswap32(d0);
        d1 = __builtin_bswap32(d1);
        d2 = __builtin_bswap32(d2);
        d3 = __builtin_bswap32(d3);
        d4 = __builtin_bswap32(d4);
      } else if (ORDER == 2) {
        d0 = (d0 << 16) | (d0 >> 16);
        d1 = (d1 << 16) | (d1 >> 16);
        d2 = (d2 << 16) | (d2 >> 16);
        d3 = (d3 << 16) | (d3 >> 16);
        d4 = (d4 << 16) | (d4 >> 16);
      }
      // drop `kb` leading stream bytes: result byte j = combined byte (7-kb-3+j)
      const uint32_t sel = 0x07060504u - 0x01010101u * kb;
      const uint32_t n0 = __builtin_amdgcn_perm(d0, d1, sel);
      const uint32_t n1 = __builtin_amdgcn_perm(d1, d2, sel);
      const uint32_t n2 = __builtin_amdgcn_perm(d2, d3, sel);
      const uint32_t n3 = __builtin_amdgcn_perm(d3, d4, sel);
      uint64_t hi = (uint64_t(n0) << 32) | n1;
      uint64_t lo = (uint64_t(n2) << 32) | n3;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s[i] = uint32_t(hi >> (64 - bps));
        hi = (hi << bps) | (lo >> (64 - bps));
        lo <<= bps;
      }
    }
    const uint32_t g = g0 + gl;
    uint16_t* dst = reinterpret_cast<uint16_t*>(out_row) + uint64_t(g) * 8;
    const uint32_t cnt = J.cols - g * 8; // samples left in the row
    if (cnt >= 8 && out_aligned) {
      uint4 o;
      o.x = s[0] | (s[1] << 16);
      o.y = s[2] | (s[3] << 16);
      o.z = s[4] | (s[5] << 16);
      o.w = s[6] | (s[7] << 16);
#if RSX_UNPACK_NT & 2
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 t;
      t.x = o.x; t.y = o.y; t.z = o.z; t.w = o.w;
      __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(dst));
#else
      *reinterpret_cast<uint4*>(dst) = o;
#endif
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (uint32_t(i) < cnt)
          dst[i] = uint16_t(s[i]);
    }
  }
}

} // namespace

size_t unpack_lds_bytes() { return size_t(SEG_CHUNKS) * 16; }

uint32_t unpack_blocks_for(uint32_t n_rows, uint32_t cols, uint32_t* segs_per_row,
                           uint32_t* groups_per_row) {
  const uint32_t groups = (cols + 7) / 8;
  const uint32_t segs = (groups + SEG_GROUPS - 1) / SEG_GROUPS;
  *segs_per_row = segs;
  *groups_per_row = groups;
  return n_rows * segs;
}

const char* unpack_kernel_name() { return "unpack_kernel"; }

hipError_t launch_unpack(int order, const UnpackJobDev* d_jobs,
                         const uint32_t* d_block_start, int n_jobs,
                         uint32_t total_blocks, const void* in_base,
                         void* out_base, hipStream_t stream) {
  if (total_blocks == 0)
    return hipSuccess;
  const dim3 grid(total_blocks), block(UNPACK_THREADS);
  const size_t lds = unpack_lds_bytes();
  const uint8_t* in = static_cast<const uint8_t*>(in_base);
  uint8_t* out = static_cast<uint8_t*>(out_base);
  switch (order) {
  case RSX_ORDER_LSB:
    hipLaunchKernelGGL(unpack_kernel<0>, grid, block, lds, stream, d_jobs,
                       d_block_start, n_jobs, in, out);
    break;
  case RSX_ORDER_MSB:
    hipLaunchKernelGGL(unpack_kernel<1>, grid, block, lds, stream, d_jobs,
                       d_block_start, n_jobs, in, out);
    break;
  case RSX_ORDER_MSB16:
    hipLaunchKernelGGL(unpack_kernel<2>, grid, block, lds, stream, d_jobs,
                       d_block_start, n_jobs, in, out);
    break;
  default:
    hipLaunchKernelGGL(unpack_kernel<3>, grid, block, lds, stream, d_jobs,
                       d_block_start, n_jobs, in, out);
    break;
  }
  return hipGetLastError();
}

} // namespace rsx
