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

// block -> job (jobs of one launch share a kernel; block ranges are prefix sums)
__device__ __forceinline__ int find_job(const uint32_t* __restrict__ job_block_start,
                                        int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  const uint32_t b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (job_block_start[mid] <= b)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// 8 samples of `bps` bits from the 20 stream bytes d0..d4 (memory order),
// starting `kb` bytes into d0.
template <int ORDER>
__device__ __forceinline__ void extract8(uint32_t d0, uint32_t d1, uint32_t d2,
                                         uint32_t d3, uint32_t d4, uint32_t kb,
                                         uint32_t bps, uint32_t (&s)[8]) {
  if (ORDER == 0) {
    // little-endian bit stream: drop `kb` low bytes, then peel from the bottom
    const uint32_t mask = (1u << bps) - 1u;
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
}

// one group of up to 8 samples: a 16-byte store when the row allows it
__device__ __forceinline__ void store8(uint16_t* __restrict__ dst, const uint32_t (&s)[8],
                                       uint32_t cnt, bool out_aligned) {
  if (cnt >= 8 && out_aligned) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t;
    t.x = s[0] | (s[1] << 16);
    t.y = s[2] | (s[3] << 16);
    t.z = s[4] | (s[5] << 16);
    t.w = s[6] | (s[7] << 16);
#if RSX_UNPACK_NT & 2
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(dst));
#else
    *reinterpret_cast<u32x4*>(dst) = t;
#endif
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (uint32_t(i) < cnt)
        dst[i] = uint16_t(s[i]);
  }
}

// ORDER: 0 LSB, 1 MSB, 2 MSB16, 3 MSB32.  POST: shift every sample right by
// J.post_shift afterwards (decode12BitRawUnpackedLeftAligned<e>,
// UncompressedDecompressor.cpp:356-378, is the 16-bit LSB / MSB walk + ">> 4").
// POST 2: every sample goes through a 256-entry u16 table instead (decode8BitRaw
// <false>); the tables of a launch sit behind its job array, post_shift = index.
template <int ORDER, int POST = 0>
__global__ __launch_bounds__(UNPACK_THREADS) void unpack_kernel(
    const UnpackJobDev* __restrict__ jobs, const uint32_t* __restrict__ job_block_start,
    int n_jobs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);

  const int job = find_job(job_block_start, n_jobs);
  const UnpackJobDev J = jobs[job];
  const uint32_t local_block = blockIdx.x - job_block_start[job];
  const uint32_t row = local_block / J.segs_per_row;
  const uint32_t seg = local_block - row * J.segs_per_row;

  const uint32_t bps = J.bps;
  const uint32_t g0 = seg * J.seg_groups; // first group of this segment
  uint32_t seg_groups = J.groups_per_row - g0;
  if (seg_groups > J.seg_groups)
    seg_groups = J.seg_groups;

  const uint8_t* __restrict__ in = in_base + J.in_offset;
  const int64_t stream_bytes = J.stream_bytes;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(in) & 15) == 0;

  // strip-relative byte range of this segment
  const int64_t start = int64_t(row) * J.in_pitch + int64_t(g0) * bps;
  uint8_t* __restrict__ out_row = out_base + J.out_offset + uint64_t(row) * J.out_pitch;
  const bool out_aligned = J.out_aligned != 0;

  // bps 8 / 16: a group is 8 / 16 whole bytes, and an LDS image of the row
  // would be read with a 2- / 4-dword lane stride (2- / 4-way bank conflicts:
  // measured 51 % / 57 % of peak against 76 % for 14 bits).  Each lane loads
  // its own group straight from HBM instead -- lanes are contiguous, so the
  // wave's loads coalesce -- with no LDS and no barrier.  Block-uniform branch.
  // (MSB16 / MSB32 chunks are relative to the strip start: their direct path
  // needs the segment to start on a chunk boundary.)
  if ((bps == 8 || bps == 16) && (ORDER <= 1 || (start & 3) == 0)) {
    uint32_t d[GROUPS_PER_THREAD][4];
#pragma unroll
    for (int k = 0; k < GROUPS_PER_THREAD; ++k) {
      const uint32_t gl = threadIdx.x + k * UNPACK_THREADS;
      d[k][0] = d[k][1] = d[k][2] = d[k][3] = 0;
      if (gl < seg_groups) {
        const int64_t off = start + int64_t(gl) * bps;
        if (off + int64_t(bps) <= stream_bytes) {
          if (bps == 16) {
            uint4 t;
            __builtin_memcpy(&t, in + off, 16);
            d[k][0] = t.x; d[k][1] = t.y; d[k][2] = t.z; d[k][3] = t.w;
          } else {
            uint2 t;
            __builtin_memcpy(&t, in + off, 8);
            d[k][0] = t.x; d[k][1] = t.y;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (uint32_t(i) < bps && off + i < stream_bytes)
              d[k][i >> 2] |= uint32_t(in[off + i]) << (8 * (i & 3));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < GROUPS_PER_THREAD; ++k) {
      const uint32_t gl = threadIdx.x + k * UNPACK_THREADS;
      if (gl >= seg_groups)
        break;
      uint32_t s[8];
      extract8<ORDER>(d[k][0], d[k][1], d[k][2], d[k][3], 0u, 0u, bps, s);
      if (POST == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          s[i] >>= J.post_shift;
      } else if (POST == 2) {
        const uint16_t* __restrict__ lut =
            reinterpret_cast<const uint16_t*>(jobs + n_jobs) + 256 * J.post_shift;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          s[i] = lut[s[i] & 0xFFu];
      }
      const uint32_t g = g0 + gl;
      store8(reinterpret_cast<uint16_t*>(out_row) + uint64_t(g) * 8, s, J.cols - g * 8,
             out_aligned);
    }
    return;
  }

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
#pragma unroll
  for (int k = 0; k < GROUPS_PER_THREAD; ++k) {
    const uint32_t gl = threadIdx.x + k * UNPACK_THREADS; // group within segment
    if (gl >= seg_groups)
      break;
    const uint32_t ob = lead + gl * bps; // LDS byte offset of the group
    const uint32_t wi = ob >> 2;
    const uint32_t kb = ob & 3; // byte shift inside the first dword
    uint32_t s[8];
    extract8<ORDER>(lds[wi], lds[wi + 1], lds[wi + 2], lds[wi + 3], lds[wi + 4], kb, bps, s);
    if (POST == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        s[i] >>= J.post_shift;
    } else if (POST == 2) {
      const uint16_t* __restrict__ lut =
          reinterpret_cast<const uint16_t*>(jobs + n_jobs) + 256 * J.post_shift;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        s[i] = lut[s[i] & 0xFFu];
    }
    const uint32_t g = g0 + gl;
    store8(reinterpret_cast<uint16_t*>(out_row) + uint64_t(g) * 8, s, J.cols - g * 8,
           out_aligned);
  }
}

// ---------------------------------------------------------------------------
// decode12BitRawWithControl<e> (UncompressedDecompressor.cpp:296-349): rows of
// `in_pitch` = 12w/8 + (w+2)/10 bytes; every 10 pixels are 15 data bytes
// followed by one control byte that is skipped.  Inside a 15-byte unit the
// pixels are a 12-bit LSB-first (little) or MSB-first (big) bit stream.
// One lane owns one 16-byte unit: one (unaligned) 16-byte load, 10 samples,
// staged through LDS so that the block emits 16-byte coalesced stores.
// J.bps carries the endianness (1 = big).
// ---------------------------------------------------------------------------
constexpr int CTRL_UNIT_PIX = 10;
constexpr int CTRL_LDS_DWORDS = UNPACK_THREADS * 5;

template <bool BIG>
__global__ __launch_bounds__(UNPACK_THREADS) void unpack_control_kernel(
    const UnpackJobDev* __restrict__ jobs, const uint32_t* __restrict__ job_block_start,
    int n_jobs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[CTRL_LDS_DWORDS];
  const int job = find_job(job_block_start, n_jobs);
  const UnpackJobDev J = jobs[job];
  const uint32_t local_block = blockIdx.x - job_block_start[job];
  const uint32_t row = local_block / J.segs_per_row;
  const uint32_t seg = local_block - row * J.segs_per_row;
  const uint32_t u0 = seg * UNPACK_THREADS;
  const uint32_t unit = u0 + threadIdx.x;

  const uint8_t* __restrict__ in_row =
      in_base + J.in_offset + uint64_t(row) * J.in_pitch;
  if (unit < J.groups_per_row) {
    const uint32_t byte0 = unit * 16;
    uint32_t d0, d1, d2, d3;
    if (byte0 + 16 <= J.in_pitch) {
      // unaligned dwordx4 load (rows are not 16-byte multiples)
      uint4 t;
      __builtin_memcpy(&t, in_row + byte0, 16);
      d0 = t.x; d1 = t.y; d2 = t.z; d3 = t.w;
    } else {
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (byte0 + i < J.in_pitch)
          w[i >> 2] |= uint32_t(in_row[byte0 + i]) << (8 * (i & 3));
      d0 = w[0]; d1 = w[1]; d2 = w[2]; d3 = w[3];
    }
    uint32_t s[CTRL_UNIT_PIX];
    if (!BIG) {
      uint64_t lo = (uint64_t(d1) << 32) | d0;
      uint64_t hi = (uint64_t(d3) << 32) | d2;
#pragma unroll
      for (int i = 0; i < CTRL_UNIT_PIX; ++i) {
        s[i] = uint32_t(lo) & 0xFFFu;
        lo = (lo >> 12) | (hi << 52);
        hi >>= 12;
      }
    } else {
      uint64_t hi = (uint64_t(__builtin_bswap32(d0)) << 32) | __builtin_bswap32(d1);
      uint64_t lo = (uint64_t(__builtin_bswap32(d2)) << 32) | __builtin_bswap32(d3);
#pragma unroll
      for (int i = 0; i < CTRL_UNIT_PIX; ++i) {
        s[i] = uint32_t(hi >> 52);
        hi = (hi << 12) | (lo >> 52);
        lo <<= 12;
      }
    }
    // 5 dwords per lane, stride 5 (odd): conflict-free
#pragma unroll
    for (int i = 0; i < 5; ++i)
      lds[threadIdx.x * 5 + i] = s[2 * i] | (s[2 * i + 1] << 16);
  }
  __syncthreads();

  // samples of this segment
  const uint32_t x0 = u0 * CTRL_UNIT_PIX;
  uint32_t n = J.cols - x0;
  if (n > UNPACK_THREADS * CTRL_UNIT_PIX)
    n = UNPACK_THREADS * CTRL_UNIT_PIX;
  uint8_t* __restrict__ out_row =
      out_base + J.out_offset + uint64_t(row) * J.out_pitch + uint64_t(x0) * 2;
  if (J.out_aligned) {
    // row starts are 16-byte aligned and x0*2 = seg*5120 is too
    const uint32_t full = n / 8;
    for (uint32_t c = threadIdx.x; c < full; c += UNPACK_THREADS) {
      const uint4 v = *reinterpret_cast<const uint4*>(&lds[c * 4]);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 t;
      t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
      __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(out_row + c * 16));
    }
    const uint16_t* l16 = reinterpret_cast<const uint16_t*>(lds);
    for (uint32_t i = full * 8 + threadIdx.x; i < n; i += UNPACK_THREADS)
      reinterpret_cast<uint16_t*>(out_row)[i] = l16[i];
  } else {
    const uint16_t* l16 = reinterpret_cast<const uint16_t*>(lds);
    for (uint32_t i = threadIdx.x; i < n; i += UNPACK_THREADS)
      reinterpret_cast<uint16_t*>(out_row)[i] = l16[i];
  }
}

// ---------------------------------------------------------------------------
// F32 images (UncompressedDecompressor.cpp:171-186, :212-245): samples of 16,
// 24 or 32 bits on byte boundaries.  One lane owns 4 samples: 8 / 12 / 16 input
// bytes (unaligned loads, contiguous across the wave), one 16-byte store.
// extendBinaryFloatingPoint<Narrow, Binary32> (common/FloatingPoint.h:109-145)
// in integer arithmetic: exact, subnormals renormalised, NaN payload kept.
// ---------------------------------------------------------------------------
template <int FRAC, int EXPW>
__device__ __forceinline__ uint32_t widen_fp(uint32_t narrow) {
  constexpr int BIAS = (1 << (EXPW - 1)) - 1;
  const uint32_t sign = (narrow >> (FRAC + EXPW)) & 1u;
  const uint32_t ne = (narrow >> FRAC) & ((1u << EXPW) - 1u);
  const uint32_t nf = narrow & ((1u << FRAC) - 1u);
  uint32_t we = ne - BIAS + 127;
  uint32_t wf = nf << (23 - FRAC);
  if (ne == (1u << EXPW) - 1u) {
    we = 255; // infinity / NaN, fraction widened
  } else if (ne == 0) {
    if (nf == 0) {
      we = 0;
      wf = 0;
    } else {
      // subnormal: normalise (shift until the hidden bit appears)
      const uint32_t sh = uint32_t(__builtin_clz(wf)) - 8u; // wf < 2^23
      we = 1 - BIAS + 127 - sh;
      wf = (wf << sh) & 0x7FFFFFu;
    }
  }
  return (sign << 31) | (we << 23) | wf;
}

// BPS in {16, 24, 32}; MSB: the bytes of a sample arrive most significant first
template <int BPS, bool MSB>
__global__ __launch_bounds__(UNPACK_THREADS) void unpack_fp_kernel(
    const UnpackJobDev* __restrict__ jobs, const uint32_t* __restrict__ job_block_start,
    int n_jobs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base) {
  const int job = find_job(job_block_start, n_jobs);
  const UnpackJobDev J = jobs[job];
  const uint32_t local_block = blockIdx.x - job_block_start[job];
  const uint32_t row = local_block / J.segs_per_row;
  const uint32_t seg = local_block - row * J.segs_per_row;
  const uint32_t g = seg * UNPACK_THREADS + threadIdx.x; // group of 4 samples
  if (g >= J.groups_per_row)
    return;
  constexpr int BYTES = BPS / 8;
  const uint8_t* __restrict__ src =
      in_base + J.in_offset + uint64_t(row) * J.in_pitch + uint64_t(g) * (4 * BYTES);
  uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(
      out_base + J.out_offset + uint64_t(row) * J.out_pitch) + uint64_t(g) * 4;
  const uint32_t cnt = J.cols - g * 4;
  uint32_t v[4] = {0, 0, 0, 0};
  if (cnt >= 4) {
    uint32_t w[4] = {0, 0, 0, 0};
    __builtin_memcpy(w, src, 4 * BYTES);
    if (BPS == 32) {
      v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
    } else if (BPS == 16) {
      v[0] = w[0] & 0xFFFFu; v[1] = w[0] >> 16; v[2] = w[1] & 0xFFFFu; v[3] = w[1] >> 16;
      if (MSB) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = ((v[i] & 0xFFu) << 8) | (v[i] >> 8);
      }
    } else {
      // 12 bytes b0..b11: sample i = bytes 3i .. 3i+2
      v[0] = w[0] & 0xFFFFFFu;
      v[1] = (w[0] >> 24) | ((w[1] & 0xFFFFu) << 8);
      v[2] = (w[1] >> 16) | ((w[2] & 0xFFu) << 16);
      v[3] = w[2] >> 8;
      if (MSB) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = ((v[i] & 0xFFu) << 16) | (v[i] & 0xFF00u) | (v[i] >> 16);
      }
    }
  } else {
    for (uint32_t i = 0; i < cnt; ++i) {
      uint32_t x = 0;
      for (int b = 0; b < BYTES; ++b) {
        const uint32_t byte = src[i * BYTES + b];
        x = MSB && BPS != 32 ? (x << 8) | byte : x | (byte << (8 * b));
      }
      v[i] = x;
    }
  }
  if (BPS == 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = widen_fp<10, 5>(v[i]);
  } else if (BPS == 24) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = widen_fp<16, 7>(v[i]);
  }
  if (cnt >= 4 && J.out_aligned) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t;
    t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(dst));
  } else {
    for (uint32_t i = 0; i < 4; ++i)
      if (i < cnt)
        dst[i] = v[i];
  }
}

} // namespace

size_t unpack_lds_bytes() { return size_t(SEG_CHUNKS) * 16; }

// A row is split into the fewest segments of <= SEG_GROUPS groups, evenly: a
// 1035-group row becomes 518 + 517, not 1024 + 11 (the near-empty trailing
// workgroups cost as much fixed work as full ones).
uint32_t unpack_blocks_for(UnpackJobDev* u) {
  const uint32_t groups = (u->cols + 7) / 8;
  const uint32_t segs = (groups + SEG_GROUPS - 1) / SEG_GROUPS;
  u->segs_per_row = segs;
  u->groups_per_row = groups;
  u->seg_groups = segs ? (groups + segs - 1) / segs : 0;
  return u->n_rows * segs;
}

uint32_t unpack_control_blocks_for(UnpackJobDev* u) {
  const uint32_t units = (u->cols + CTRL_UNIT_PIX - 1) / CTRL_UNIT_PIX;
  const uint32_t segs = (units + UNPACK_THREADS - 1) / UNPACK_THREADS;
  u->segs_per_row = segs;
  u->groups_per_row = units;
  u->seg_groups = UNPACK_THREADS;
  return u->n_rows * segs;
}

uint32_t unpack_fp_blocks_for(UnpackJobDev* u) {
  const uint32_t groups = (u->cols + 3) / 4;
  const uint32_t segs = (groups + UNPACK_THREADS - 1) / UNPACK_THREADS;
  u->segs_per_row = segs;
  u->groups_per_row = groups;
  u->seg_groups = UNPACK_THREADS;
  return u->n_rows * segs;
}

const char* unpack_kernel_name() { return "unpack_kernel"; }

hipError_t launch_unpack_mode(int mode, int order, const UnpackJobDev* d_jobs,
                              const uint32_t* d_block_start, int n_jobs,
                              uint32_t total_blocks, const void* in_base,
                              void* out_base, hipStream_t stream) {
  if (mode == UNPACK_MODE_PACKED)
    return launch_unpack(order, d_jobs, d_block_start, n_jobs, total_blocks, in_base,
                         out_base, stream);
  if (total_blocks == 0)
    return hipSuccess;
  const dim3 grid(total_blocks), block(UNPACK_THREADS);
  const uint8_t* in = static_cast<const uint8_t*>(in_base);
  uint8_t* out = static_cast<uint8_t*>(out_base);
  if (mode == UNPACK_MODE_FP) {
    // FP launch classes are (byte order, bit width): `order` = width << 8 | order
    const int bps = order >> 8;
    const bool msb = (order & 0xFF) == RSX_ORDER_MSB;
    if (bps == 32)
      hipLaunchKernelGGL((unpack_fp_kernel<32, false>), grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else if (bps == 16 && msb)
      hipLaunchKernelGGL((unpack_fp_kernel<16, true>), grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else if (bps == 16)
      hipLaunchKernelGGL((unpack_fp_kernel<16, false>), grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else if (msb)
      hipLaunchKernelGGL((unpack_fp_kernel<24, true>), grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else
      hipLaunchKernelGGL((unpack_fp_kernel<24, false>), grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    return hipGetLastError();
  }
  if (mode == UNPACK_MODE_LUT8) {
    hipLaunchKernelGGL((unpack_kernel<0, 2>), grid, block, unpack_lds_bytes(), stream,
                       d_jobs, d_block_start, n_jobs, in, out);
    return hipGetLastError();
  }
  if (mode == UNPACK_MODE_SHIFT) {
    const size_t lds = unpack_lds_bytes();
    if (order == RSX_ORDER_LSB)
      hipLaunchKernelGGL((unpack_kernel<0, 1>), grid, block, lds, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else
      hipLaunchKernelGGL((unpack_kernel<1, 1>), grid, block, lds, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
  } else {
    if (order == RSX_ORDER_LSB)
      hipLaunchKernelGGL(unpack_control_kernel<false>, grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
    else
      hipLaunchKernelGGL(unpack_control_kernel<true>, grid, block, 0, stream, d_jobs,
                         d_block_start, n_jobs, in, out);
  }
  return hipGetLastError();
}

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

// ---------------------------------------------------------------------------
// Measurement aid (rsx_probe_stream_copy): the plainest kernel that moves the
// same bytes as an unpack launch -- 16-byte loads of `n_in16` chunks, 16-byte
// non-temporal stores of `n_out16` chunks, nothing in between.  Its rate is the
// copy ceiling of this device for that read:write mix; bench.py reports the
// unpack kernel against it next to the 8 TB/s vendor peak.
// ---------------------------------------------------------------------------
namespace {
template <int U>
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint4* __restrict__ in,
                                                           uint64_t n_in16,
                                                           uint4* __restrict__ out,
                                                           uint64_t n_out16) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint64_t n = n_in16 > n_out16 ? n_in16 : n_out16;
  // a workgroup owns U consecutive 4 KB pieces; all loads are issued before the stores
  const uint64_t base = (uint64_t(blockIdx.x) * U) * 256 + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t i = base + uint64_t(u) * 256;
    v[u] = u32x4{0u, 0u, 0u, 0u};
    if (i < n_in16)
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + i);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t i = base + uint64_t(u) * 256;
    if (i < n_out16)
      __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(out) + i);
    else if (i < n && v[u].x == 0x12345678u && v[u].y == 0x9ABCDEF0u && v[u].z == v[u].w)
      out[0].x = v[u].w; // keeps the load alive when there is no matching store
  }
}
} // namespace

hipError_t launch_stream_probe(const void* in, uint64_t in_bytes, void* out,
                               uint64_t out_bytes, hipStream_t stream) {
  const uint64_t n_in16 = in_bytes / 16, n_out16 = out_bytes / 16;
  const uint64_t n = n_in16 > n_out16 ? n_in16 : n_out16;
  if (n == 0)
    return hipSuccess;
  // one chunk per lane, like the unpack kernels (RSX_PROBE_UNROLL = 2 / 4: more
  // loads in flight per lane)
  static const int unroll = getenv("RSX_PROBE_UNROLL") ? atoi(getenv("RSX_PROBE_UNROLL")) : 1;
  const int U = unroll == 4 ? 4 : unroll == 2 ? 2 : 1;
  const uint64_t blocks = (n + 256ull * U - 1) / (256ull * U);
  const dim3 grid{uint32_t(blocks)}, block(256);
  const uint4* pi = static_cast<const uint4*>(in);
  uint4* po = static_cast<uint4*>(out);
  if (U == 4)
    hipLaunchKernelGGL(stream_probe_kernel<4>, grid, block, 0, stream, pi, n_in16, po, n_out16);
  else if (U == 2)
    hipLaunchKernelGGL(stream_probe_kernel<2>, grid, block, 0, stream, pi, n_in16, po, n_out16);
  else
    hipLaunchKernelGGL(stream_probe_kernel<1>, grid, block, 0, stream, pi, n_in16, po, n_out16);
  return hipGetLastError();
}

} // namespace rsx
