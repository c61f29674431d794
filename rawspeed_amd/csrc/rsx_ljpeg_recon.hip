// rsx_ljpeg_recon.hip -- reconstruction half of the lossless-JPEG family
// pipeline: the entropy kernels (rsx_ljpeg.hip) leave stream-ordered int16
// differences in scratch; these kernels turn them into pixels.
//   K5  lj_vseed_kernel          predictor seeds of every stream row
//   K6  lj_predict_kernel<N,P>   row scans + output mapping (generic)
//       lj_predict_fast_kernel   2 / 4 interleaved components (the BASELINE shapes)
//   nk_vseed / nk_predict        NikonDecompressor / PentaxDecompressor (int sums,
//                                clamp or range check, curve + dither)
#include "rsx_ljpeg_dev.h"

namespace rsx {

namespace {

// final pixels are written once and not read again by this pipeline: bypass the
// caches (measured +18 % on the sRaw kernel, +6..10 % on the unpack kernel)
__device__ __forceinline__ void store_nt16(void* p, uint4 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, static_cast<u32x4*>(p));
}
// the differences are read exactly once (A/B: -2 % on the cfg 3 pipeline)
__device__ __forceinline__ uint4 load_diffs16(const int16_t* p) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(t.x, t.y, t.z, t.w);
}

// ---------------------------------------------------------------------------
// K5: predictor seeds of the stream rows
//   seed(r, c) = init_pred[c] + sum_{r' < r} D[r'][seed_pos[c]]   (mod 2^16)
// seed_pos[c] = c for interleaved components; (0, gs-2, gs-1) for Canon sRaw
// groups (Cr2DecompressorImpl.h:443-444).
// ---------------------------------------------------------------------------
// 1024 lanes per stream, one row per lane per step (the per-row reads are 8-byte
// gathers at a pitch of a whole stream row, so they are issued for many rows at
// once); block-wide exclusive scan per step with shuffles, carry across steps.
__global__ __launch_bounds__(VS_T) void lj_vseed_kernel(LjArgs a) {
  __shared__ uint32_t wtot[VS_T / 64][4];
  __shared__ uint32_t carry_s[4];
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (a.results[s].status != 0 || S.kind == 2)
    return;
  if (!lj_recon_takes(a, s, S))
    return; // reconstructed by the fused decode (rsx_ljpeg_direct.hip)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (the stream record is read once: stores to V could alias it for the compiler)
  const uint32_t rows = S.rows, N = S.n_comp, row_samples = S.row_samples;
  const bool no_vertical = S.no_vertical != 0;
  uint32_t seed_pos[4], init_pred[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    seed_pos[c] = S.seed_pos[c];
    init_pred[c] = S.init_pred[c];
  }
  const int16_t* __restrict__ D = a.diffs + S.diff_offset;
  uint16_t* __restrict__ V = a.vseed + uint64_t(S.first_row) * 4;
  if (tid < 4)
    carry_s[tid] = tid < int(N) ? init_pred[tid] : 0u;
  __syncthreads();
  // the gathers of the next step are in flight while this one is scanned
  auto gather = [&](uint32_t r, uint32_t (&d)[4]) {
    d[0] = d[1] = d[2] = d[3] = 0;
    if (r < rows) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (uint32_t(c) < N)
          d[c] = uint32_t(int32_t(D[uint64_t(r) * row_samples + seed_pos[c]]));
    }
  };
  uint32_t dn[4];
  gather(tid, dn);
  for (uint32_t r0 = 0; r0 < rows; r0 += VS_T) {
    const uint32_t r = r0 + tid;
    uint32_t d[4] = {dn[0], dn[1], dn[2], dn[3]};
    gather(r + VS_T, dn);
    uint32_t inc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t x = d[c];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
          x += y;
      }
      inc[c] = x;
      if (lane == 63)
        wtot[wv][c] = x;
    }
    __syncthreads();
    uint32_t base[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint32_t o = carry_s[c];
      for (int w = 0; w < wv; ++w)
        o += wtot[w][c];
      base[c] = o;
    }
    if (r < rows) {
#pragma unroll
      for (int c = 0; c < 4; ++c) // exclusive; Hasselblad rows all start from initPred
        if (uint32_t(c) < N)
          V[uint64_t(r) * 4 + c] =
              no_vertical ? uint16_t(init_pred[c]) : uint16_t(base[c] + inc[c] - d[c]);
    }
    __syncthreads();
    if (tid == VS_T - 1)
      for (int c = 0; c < 4; ++c)
        carry_s[c] = base[c] + inc[c];
    __syncthreads();
  }
}

// offset inside a strip -> (row, column).  Strips of real images hold fewer than
// 2^32 samples, and 64-bit division is an order of magnitude more instructions
// than the 32-bit one on this hardware.
__device__ __forceinline__ void strip_divmod(uint64_t off, uint32_t w, uint32_t* row,
                                             uint32_t* col) {
  if ((off >> 32) == 0) {
    const uint32_t o = uint32_t(off), q = o / w;
    *row = q;
    *col = o - q * w;
  } else {
    *row = uint32_t(off / w);
    *col = uint32_t(off % w);
  }
}

// ---------------------------------------------------------------------------
// K6: row reconstruction + output mapping, one wavefront per stream row
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lj_store_sample(const LjArgs& a, const LjStreamDev& S,
                                                uint32_t r, uint32_t sidx,
                                                uint16_t val) {
  uint8_t* img = a.out_base + S.img_offset;
  if (S.kind == 0) {
    const uint32_t m = sidx / S.n_comp, c = sidx - m * S.n_comp;
    const uint32_t col = S.mcu_w * m + (c % S.mcu_w);
    if (col >= S.keep_samples)
      return;
    const uint32_t row = S.out_y + S.mcu_h * r + c / S.mcu_w;
    reinterpret_cast<uint16_t*>(img + uint64_t(row) * S.img_pitch)[S.out_x + col] = val;
  } else {
    const uint64_t k = uint64_t(r) * S.row_samples + sidx;
    const Cr2Strip* st = a.strips + S.strip_base;
    uint32_t q = 0;
    while (q + 1 < S.n_strips && k >= st[q + 1].first_sample)
      ++q;
    uint32_t srow, scol;
    strip_divmod(k - st[q].first_sample, st[q].w, &srow, &scol);
    reinterpret_cast<uint16_t*>(img + uint64_t(st[q].y0 + srow) * S.img_pitch)[st[q].x0 + scol] =
        val;
  }
}

// P == N: sample s belongs to component s % N.  P != N (N == 3): Canon sRaw
// groups of P samples, P - 2 luma samples (component 0) then Cb, Cr
// (Cr2DecompressorImpl.h:455-462).
template <int N, int P = N>
__global__ __launch_bounds__(LJ_T) void lj_predict_kernel(LjArgs a) {
  constexpr bool DYN = (8 % P) != 0; // lane chunks of 8 do not start on a group
  auto comp_of = [](int ph) -> int {
    return P == N ? ph : (ph < P - 2 ? 0 : ph - (P - 3));
  };
  const uint32_t grow = blockIdx.x * (LJ_T / 64) + (threadIdx.x >> 6);
  if (grow >= a.total_rows)
    return;
  const int lane = threadIdx.x & 63;
  // row -> stream (first_row is increasing)
  uint32_t lo = 0, hi = a.n_streams - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.streams[mid].first_row <= grow)
      lo = mid;
    else
      hi = mid - 1;
  }
  const LjStreamDev& S = a.streams[lo];
  if (int(S.n_comp) != N || int(S.period) != P || S.kind == 2 ||
      a.results[lo].status != 0)
    return;
  if (!lj_recon_takes(a, lo, S))
    return;
  const uint32_t r = grow - S.first_row;
  if (r >= S.rows)
    return;
  const uint64_t row0 = uint64_t(r) * S.row_samples;
  uint32_t n = S.scan_samples;
  const uint64_t n_diffs = S.pair ? 2 * S.needed : S.needed; // a pair symbol = 2 differences
  if (row0 + n > n_diffs)
    n = uint32_t(n_diffs - row0);
  const int16_t* __restrict__ D = a.diffs + S.diff_offset + row0;
  const bool in_aligned = ((S.diff_offset + row0) & 7) == 0;
  uint32_t carry[N];
#pragma unroll
  for (int c = 0; c < N; ++c)
    carry[c] = a.vseed[(uint64_t(S.first_row) + r) * 4 + c];

  // 8 differences of this lane for the step starting at q0 (packed 2 x u16 per
  // dword); loads run two steps ahead of the scan so that HBM latency overlaps
  auto load8 = [&](uint32_t q0) -> uint4 {
    const uint32_t q = q0 + lane * 8;
    if (q + 8 <= n && in_aligned)
      return *reinterpret_cast<const uint4*>(D + q);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (q + i < n)
        w[i >> 1] |= uint32_t(uint16_t(D[q + i])) << (16 * (i & 1));
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  uint4 t0 = load8(0);
  uint4 t1 = 512 < n ? load8(512) : make_uint4(0, 0, 0, 0);
  for (uint32_t q0 = 0; q0 < n; q0 += 512) {
    const uint32_t q = q0 + lane * 8;
    const uint4 t = t0;
    t0 = t1;
    if (q0 + 1024 < n)
      t1 = load8(q0 + 1024);
    uint32_t v[8];
    v[0] = t.x & 0xFFFF; v[1] = t.x >> 16; v[2] = t.y & 0xFFFF; v[3] = t.y >> 16;
    v[4] = t.z & 0xFFFF; v[5] = t.z >> 16; v[6] = t.w & 0xFFFF; v[7] = t.w >> 16;
    // component of v[i] is comp_of((q + i) % P); rot = q % P (0 unless DYN)
    const int rot = DYN ? int(q % P) : 0;
    uint32_t run[N];
#pragma unroll
    for (int c = 0; c < N; ++c)
      run[c] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (DYN) {
        // select by component without dynamic register indexing
        const int c = comp_of((rot + i) % P);
        uint32_t t = (c == 0 ? run[0] : (c == 1 ? run[1 % N] : run[2 % N])) + v[i];
        if (c == 0) run[0] = t; else if (c == 1) run[1 % N] = t; else run[2 % N] = t;
        v[i] = t;
      } else {
        const int cs = comp_of(i % P);
        run[cs] += v[i];
        v[i] = run[cs];
      }
    }
    // exclusive wave scan of the lane totals, per component
    uint32_t excl[N], tot[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      uint32_t x = run[c];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
          x += y;
      }
      tot[c] = __shfl(x, 63, 64);
      excl[c] = x - run[c] + carry[c];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (DYN) {
        const int c = comp_of((rot + i) % P);
        v[i] += (c == 0 ? excl[0] : (c == 1 ? excl[1 % N] : excl[2 % N]));
      } else {
        v[i] += excl[comp_of(i % P)];
      }
      v[i] &= 0xFFFFu;
    }
#pragma unroll
    for (int c = 0; c < N; ++c)
      carry[c] += tot[c];

    // ---- output -----------------------------------------------------------
    if (q >= n)
      continue;
    bool done = false;
    if (q + 8 <= n) {
      uint8_t* img = a.out_base + S.img_offset;
      uint16_t* p = nullptr;
      if (S.kind == 0 && S.mcu_h == 1) {
        if (q + 8 <= S.keep_samples)
          p = reinterpret_cast<uint16_t*>(img + uint64_t(S.out_y + r) * S.img_pitch) +
              S.out_x + q;
      } else if (S.kind == 1) {
        const uint64_t k = row0 + q;
        const Cr2Strip* st = a.strips + S.strip_base;
        uint32_t z = 0;
        while (z + 1 < S.n_strips && k >= st[z + 1].first_sample)
          ++z;
        uint32_t srow, col;
        strip_divmod(k - st[z].first_sample, st[z].w, &srow, &col);
        if (col + 8 <= st[z].w)
          p = reinterpret_cast<uint16_t*>(img + uint64_t(st[z].y0 + srow) * S.img_pitch) +
              st[z].x0 + col;
      }
      if (p && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        uint4 o;
        o.x = v[0] | (v[1] << 16);
        o.y = v[2] | (v[3] << 16);
        o.z = v[4] | (v[5] << 16);
        o.w = v[6] | (v[7] << 16);
        store_nt16(p, o);
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (q + i < n)
          lj_store_sample(a, S, r, q + i, uint16_t(v[i]));
    }
  }
}

// ---------------------------------------------------------------------------
// K6 fast path (2 or 4 components, the BASELINE shapes).  One wavefront per
// stream row.  The row is walked in chunks of 2048 samples: coalesced 16-byte
// loads -> LDS transpose so that every lane owns 32 CONSECUTIVE samples -> the
// lane scans them with packed 16-bit adds (both components of a pair at once)
// -> one DPP wave scan of the 64 lane totals (row_shr / row_bcast, no LDS
// round trips) -> back through LDS to the coalesced layout -> output mapping.
// ---------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t x, uint32_t y) {
  const u16x2 r = __builtin_bit_cast(u16x2, x) + __builtin_bit_cast(u16x2, y);
  return __builtin_bit_cast(uint32_t, r);
}

// inclusive wave64 scan with packed 16-bit adds (DPP: rows of 16, then row
// broadcasts -- gfx9 encodings row_shr:n = 0x110+n, row_bcast15 = 0x142,
// row_bcast31 = 0x143)
__device__ __forceinline__ uint32_t pk_wave_scan(uint32_t x) {
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, false)));
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, false)));
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, false)));
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, false)));
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false)));
  x = pk_add(x, uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false)));
  return x;
}

// store the 8 reconstructed samples that start at sample q of stream row r
__device__ __forceinline__ void lj_store8(const LjArgs& a, const LjStreamDev& S, uint32_t r,
                                          uint64_t row0, uint32_t q, uint32_t n,
                                          const uint4& o) {
  if (q >= n)
    return;
  if (q + 8 <= n) {
    uint8_t* img = a.out_base + S.img_offset;
    uint16_t* p = nullptr;
    if (S.kind == 0 && S.mcu_h == 1) {
      if (q + 8 <= S.keep_samples)
        p = reinterpret_cast<uint16_t*>(img + uint64_t(S.out_y + r) * S.img_pitch) +
            S.out_x + q;
    } else if (S.kind == 1) {
      const uint64_t k = row0 + q;
      const Cr2Strip* st = a.strips + S.strip_base;
      uint32_t z = 0;
      while (z + 1 < S.n_strips && k >= st[z + 1].first_sample)
        ++z;
      uint32_t srow, col;
      strip_divmod(k - st[z].first_sample, st[z].w, &srow, &col);
      if (col + 8 <= st[z].w)
        p = reinterpret_cast<uint16_t*>(img + uint64_t(st[z].y0 + srow) * S.img_pitch) +
            st[z].x0 + col;
    }
    if (p && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      store_nt16(p, o);
      return;
    }
  }
  const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (q + i < n)
      lj_store_sample(a, S, r, q + i, uint16_t(w[i >> 1] >> (16 * (i & 1))));
}

constexpr int PF_STRIDE = 80;   // LDS bytes per lane (64 + 16 pad against bank conflicts)
constexpr int PF_CHUNK = 2048;  // samples per wave per step

template <int N>
__global__ __launch_bounds__(LJ_T) void lj_predict_fast_kernel(LjArgs a) {
  static_assert(N == 2 || N == 4, "fast path handles 2 or 4 components");
  __shared__ __attribute__((aligned(16))) uint8_t tr_all[LJ_T / 64][64 * PF_STRIDE];
  const uint32_t grow = blockIdx.x * (LJ_T / 64) + (threadIdx.x >> 6);
  if (grow >= a.total_rows)
    return;
  const int lane = threadIdx.x & 63;
  uint8_t* tr = tr_all[threadIdx.x >> 6];
  uint32_t lo = 0, hi = a.n_streams - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.streams[mid].first_row <= grow)
      lo = mid;
    else
      hi = mid - 1;
  }
  const LjStreamDev& S = a.streams[lo];
  if (int(S.n_comp) != N || int(S.period) != N || S.kind == 2 ||
      a.results[lo].status != 0)
    return;
  if (!lj_recon_takes(a, lo, S))
    return;
  const uint32_t r = grow - S.first_row;
  if (r >= S.rows)
    return;
  const uint64_t row0 = uint64_t(r) * S.row_samples;
  uint32_t n = S.scan_samples;
  const uint64_t n_diffs = S.pair ? 2 * S.needed : S.needed; // a pair symbol = 2 differences
  if (row0 + n > n_diffs)
    n = uint32_t(n_diffs - row0);
  const int16_t* __restrict__ D = a.diffs + S.diff_offset + row0;
  const bool in_aligned = ((S.diff_offset + row0) & 7) == 0;
  // running predictor, packed pairs: (c0,c1) [, (c2,c3)]
  const uint16_t* vs = a.vseed + (uint64_t(S.first_row) + r) * 4;
  uint32_t carry0 = uint32_t(vs[0]) | (uint32_t(vs[1]) << 16);
  uint32_t carry1 = N == 4 ? (uint32_t(vs[2]) | (uint32_t(vs[3]) << 16)) : 0u;

  auto load8 = [&](uint32_t q) -> uint4 {
    if (q + 8 <= n && in_aligned)
      return load_diffs16(D + q);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (q + i < n)
        w[i >> 1] |= uint32_t(uint16_t(D[q + i])) << (16 * (i & 1));
    return make_uint4(w[0], w[1], w[2], w[3]);
  };

  uint4 nx[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
    nx[m] = load8((m * 64 + lane) * 8);
  for (uint32_t c0 = 0; c0 < n; c0 += PF_CHUNK) {
    // coalesced registers -> LDS (lane g/4 owns uint4 g)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int g = m * 64 + lane;
      *reinterpret_cast<uint4*>(tr + (g >> 2) * PF_STRIDE + (g & 3) * 16) = nx[m];
    }
    // prefetch the next chunk while this one is processed
    if (c0 + PF_CHUNK < n) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        nx[m] = load8(c0 + PF_CHUNK + (m * 64 + lane) * 8);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 t = *reinterpret_cast<const uint4*>(tr + lane * PF_STRIDE + i * 16);
      w[4 * i] = t.x;
      w[4 * i + 1] = t.y;
      w[4 * i + 2] = t.z;
      w[4 * i + 3] = t.w;
    }
    // lane-local inclusive scan (dword = one (c0,c1) pair; N == 4: pairs alternate)
#pragma unroll
    for (int i = N / 2; i < 16; ++i)
      w[i] = pk_add(w[i], w[i - N / 2]);
    // wave scan of the lane totals
    const uint32_t inc0 = pk_wave_scan(w[N == 2 ? 15 : 14]);
    const uint32_t inc1 = N == 4 ? pk_wave_scan(w[15]) : 0u;
    // exclusive offset of this lane = inclusive of the lane before + carry
    uint32_t ex0 = uint32_t(__builtin_amdgcn_update_dpp(0, int(inc0), 0x138 /*wave_shr:1*/, 0xF, 0xF, false));
    uint32_t ex1 = N == 4 ? uint32_t(__builtin_amdgcn_update_dpp(0, int(inc1), 0x138, 0xF, 0xF, false)) : 0u;
    ex0 = pk_add(ex0, carry0);
    ex1 = pk_add(ex1, carry1);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      w[i] = pk_add(w[i], (N == 2 || (i & 1) == 0) ? ex0 : ex1);
    carry0 = pk_add(carry0, uint32_t(__builtin_amdgcn_readlane(int(inc0), 63)));
    if (N == 4)
      carry1 = pk_add(carry1, uint32_t(__builtin_amdgcn_readlane(int(inc1), 63)));
    // back to the coalesced layout
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<uint4*>(tr + lane * PF_STRIDE + i * 16) =
          make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int g = m * 64 + lane;
      const uint4 o = *reinterpret_cast<const uint4*>(tr + (g >> 2) * PF_STRIDE + (g & 3) * 16);
      lj_store8(a, S, r, row0, c0 + uint32_t(g) * 8, n, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// NikonDecompressor reconstruction (NikonDecompressor.cpp:515-539).  Unlike the
// JPEG predictors these sums are plain ints -- nothing wraps mod 2^16 -- and
// only the stored value is clamped to 15 bits.
//   pred(y, x) = pUp_y[x & 1] + sum_{x' <= x, x' = x (2)} D[y][x']
//   pUp_y[c]   = pUp_init[y & 1][c] + sum_{y' < y, y' = y (2)} D[y'][c]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(VS_T) void nk_vseed_kernel(LjArgs a) {
  __shared__ int32_t wtot[VS_T / 64][4];
  __shared__ int32_t carry_s[4];
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (S.kind != 2 || a.results[s].status != 0 || lj_recon_skips_given_up(a, s, S))
    return;
  const NkStreamDev& K = a.nk[s];
  if (K.sony)
    return; // sony_* kernels
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t rows = S.rows;
  const int16_t* __restrict__ D = a.diffs + S.diff_offset;
  int32_t* __restrict__ V = reinterpret_cast<int32_t*>(a.vseed) + uint64_t(S.first_row) * 2;
  if (tid < 4)
    carry_s[tid] = K.pup_in ? K.pup_in[tid] : K.p_up[tid];
  __syncthreads();
  for (uint32_t r0 = 0; r0 < rows; r0 += VS_T) {
    const uint32_t r = r0 + tid;
    const uint32_t par = (S.out_y + r) & 1u;
    int32_t d[4] = {0, 0, 0, 0};
    if (r < rows) {
      d[2 * par] = D[uint64_t(r) * S.row_samples];
      d[2 * par + 1] = D[uint64_t(r) * S.row_samples + 1];
    }
    int32_t inc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int32_t x = d[c];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t y = __shfl_up(x, o, 64);
        if (lane >= o)
          x += y;
      }
      inc[c] = x;
      if (lane == 63)
        wtot[wv][c] = x;
    }
    __syncthreads();
    int32_t base[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int32_t o = carry_s[c];
      for (int w = 0; w < wv; ++w)
        o += wtot[w][c];
      base[c] = o;
    }
    if (r < rows) {
      // exclusive: the value of pUp[row & 1] when row r starts
      V[uint64_t(r) * 2] = par ? base[2] + inc[2] - d[2] : base[0] + inc[0] - d[0];
      V[uint64_t(r) * 2 + 1] = par ? base[3] + inc[3] - d[3] : base[1] + inc[1] - d[1];
    }
    __syncthreads();
    if (tid == VS_T - 1)
      for (int c = 0; c < 4; ++c)
        carry_s[c] = base[c] + inc[c];
    __syncthreads();
  }
  if (tid < 4)
    a.nk_pup[s * 4 + tid] = carry_s[tid];
}

// The dither state of RawImageDataU16::setWithLookUp (common/RawImage.h:335-353)
// is a lag-1 multiply-with-carry generator, r' = 15700 * (r & 65535) + (r >> 16),
// advanced once per pixel in decode order.  With m = 15700 * 2^16 - 1 one has
// 2^16 * r' = r (mod m), i.e. r_n = r_0 * 15700^n mod m for r_0 < m (the seed is
// 24 bits): a lane jumps to its first pixel and then steps like the reference.
constexpr uint64_t NK_MWC_A = 15700, NK_MWC_M = NK_MWC_A * 65536 - 1;
__device__ __forceinline__ uint32_t nk_mulmod(uint32_t x, uint32_t y) {
  return uint32_t((uint64_t(x) * y) % NK_MWC_M);
}

__global__ __launch_bounds__(LJ_T) void nk_predict_kernel(LjArgs a) {
  const uint32_t grow = blockIdx.x * (LJ_T / 64) + (threadIdx.x >> 6);
  if (grow >= a.total_rows)
    return;
  const int lane = threadIdx.x & 63;
  uint32_t lo = 0, hi = a.n_streams - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.streams[mid].first_row <= grow)
      lo = mid;
    else
      hi = mid - 1;
  }
  const LjStreamDev& S = a.streams[lo];
  if (S.kind != 2 || a.results[lo].status != 0 || lj_recon_skips_given_up(a, lo, S))
    return;
  const NkStreamDev& K = a.nk[lo];
  if (K.sony)
    return; // sony_predict_kernel
  const uint32_t r = grow - S.first_row;
  if (r >= S.rows)
    return;
  const uint32_t W = S.row_samples;
  const uint32_t y = S.out_y + r;
  const uint64_t row0 = uint64_t(r) * W;
  const int16_t* __restrict__ D = a.diffs + S.diff_offset + row0;
  const bool in_aligned = ((S.diff_offset + row0) & 7) == 0;
  const int32_t* V = reinterpret_cast<const int32_t*>(a.vseed) + uint64_t(grow) * 2;
  int32_t carry0 = V[0], carry1 = V[1];
  uint8_t* out_row = a.out_base + S.img_offset + uint64_t(y) * S.img_pitch;
  const bool out_aligned = (reinterpret_cast<uintptr_t>(out_row) & 15) == 0;

  const bool dither = K.uncorrected == 0;
  const uint32_t* __restrict__ tab = a.nk_tables + K.table_off;
  uint32_t step_state = 0, lane_mul = 1, a512 = 1;
  if (dither) {
    const uint8_t* in0 = a.in_base + K.seed_offset;
    const uint32_t seed = (uint32_t(in0[0]) << 16) | (uint32_t(in0[1]) << 8) | in0[2];
    step_state = nk_mulmod(seed, a.nk_rowpow[K.rowpow_off + r]);
    // 15700^(8 * lane) and 15700^512 by square-and-multiply (once per row)
    uint32_t b = uint32_t(NK_MWC_A), e = 8u * uint32_t(lane);
    for (int i = 0; i < 9; ++i) {
      if (e & 1u)
        lane_mul = nk_mulmod(lane_mul, b);
      b = nk_mulmod(b, b);
      e >>= 1;
    }
    a512 = b; // b = 15700^(2^9)
  }

  // 8 differences of this lane for the step starting at q0 (2 x int16 per dword);
  // loads run two steps ahead of the scan so that HBM latency overlaps
  auto load8 = [&](uint32_t q0) -> uint4 {
    const uint32_t q = q0 + lane * 8;
    if (q + 8 <= W && in_aligned)
      return *reinterpret_cast<const uint4*>(D + q);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (q + i < W)
        w[i >> 1] |= uint32_t(uint16_t(D[q + i])) << (16 * (i & 1));
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  uint4 t0 = load8(0);
  uint4 t1 = 512 < W ? load8(512) : make_uint4(0, 0, 0, 0);
  for (uint32_t q0 = 0; q0 < W; q0 += 512) {
    const uint32_t q = q0 + lane * 8;
    const uint4 t = t0;
    t0 = t1;
    if (q0 + 1024 < W)
      t1 = load8(q0 + 1024);
    int32_t v[8];
    v[0] = int16_t(t.x); v[1] = int32_t(t.x) >> 16;
    v[2] = int16_t(t.y); v[3] = int32_t(t.y) >> 16;
    v[4] = int16_t(t.z); v[5] = int32_t(t.z) >> 16;
    v[6] = int16_t(t.w); v[7] = int32_t(t.w) >> 16;
    int32_t run0 = 0, run1 = 0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      run0 += v[i];
      v[i] = run0;
      run1 += v[i + 1];
      v[i + 1] = run1;
    }
    int32_t x0 = run0, x1 = run1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y0 = __shfl_up(x0, o, 64), y1 = __shfl_up(x1, o, 64);
      if (lane >= o) {
        x0 += y0;
        x1 += y1;
      }
    }
    const int32_t tot0 = __shfl(x0, 63, 64), tot1 = __shfl(x1, 63, 64);
    const int32_t e0 = x0 - run0 + carry0, e1 = x1 - run1 + carry1;
    carry0 += tot0;
    carry1 += tot1;

    uint32_t st = 0;
    if (dither) {
      st = nk_mulmod(step_state, lane_mul);
      step_state = nk_mulmod(step_state, a512);
    }
    uint32_t px[8];
    if (K.pentax) {
      // isIntN(value, bits) (PentaxDecompressor.cpp:170, SamsungV1Decompressor.cpp:
      // 133): the value as unsigned must fit `bits` bits
      bool bad = false;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        bad |= q + i < W && (uint32_t(v[i] + ((i & 1) ? e1 : e0)) >> K.pentax) != 0;
      if (bad)
        atomicCAS(&a.results[lo].status, 0u, uint32_t(RSX_ERR_VALUE_RANGE));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int32_t p = v[i] + ((i & 1) ? e1 : e0);
      if (K.pentax)
        p &= 0xFFFF;
      else
        p = p < 0 ? 0 : (p > 32767 ? 32767 : p); // clampBits(pred, 15)
      if (dither) {
        const uint32_t t = tab[p];
        px[i] = ((t & 0xFFFFu) + (((t >> 16) * (st & 2047u) + 1024u) >> 12)) & 0xFFFFu;
        st = 15700u * (st & 65535u) + (st >> 16);
      } else {
        px[i] = uint32_t(p);
      }
    }
    if (q >= W)
      continue;
    uint16_t* dst = reinterpret_cast<uint16_t*>(out_row) + q;
    if (q + 8 <= W && out_aligned) {
      uint4 o;
      o.x = px[0] | (px[1] << 16);
      o.y = px[2] | (px[3] << 16);
      o.z = px[4] | (px[5] << 16);
      o.w = px[6] | (px[7] << 16);
      store_nt16(dst, o);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (q + i < W)
          dst[i] = uint16_t(px[i]);
    }
  }
}


// ---------------------------------------------------------------------------
// SonyArw1Decompressor (decompressors/SonyArw1Decompressor.cpp:59-93).  A stream
// row is an image column (the rightmost first): its H samples are the even image
// rows top to bottom, then the odd ones.  ONE predictor, starting at 0, runs
// through all columns, so the seed of stream row r is the sum of every difference
// of the rows before it:
//   sony_rowsum  one wavefront per stream row: its total            -> V[2r+1]
//   sony_vseed   one workgroup per stream: exclusive scan of totals -> V[2r]
//   sony_predict one wavefront per stream row: int32 scan, the 0..4095 range
//                check (isIntN(pred, 12), adt/Bit.h:85-90), values back in place
//   sony_transpose  64 x 64 tiles: stream order -> image columns
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool sony_row(const LjArgs& a, uint32_t grow, uint32_t* stream,
                                         uint32_t* row) {
  if (grow >= a.total_rows)
    return false;
  uint32_t lo = 0, hi = a.n_streams - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.streams[mid].first_row <= grow)
      lo = mid;
    else
      hi = mid - 1;
  }
  const LjStreamDev& S = a.streams[lo];
  if (S.kind != 2 || !a.nk[lo].sony || a.results[lo].status != 0)
    return false;
  const uint32_t r = grow - S.first_row;
  if (r >= S.rows)
    return false;
  *stream = lo;
  *row = r;
  return true;
}

__device__ __forceinline__ uint4 sony_load8(const int16_t* __restrict__ D, uint32_t q,
                                            uint32_t n, bool aligned) {
  if (q + 8 <= n && aligned)
    return *reinterpret_cast<const uint4*>(D + q);
  uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (q + i < n)
      w[i >> 1] |= uint32_t(uint16_t(D[q + i])) << (16 * (i & 1));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(LJ_T) void sony_rowsum_kernel(LjArgs a) {
  const uint32_t grow = blockIdx.x * (LJ_T / 64) + (threadIdx.x >> 6);
  uint32_t s, r;
  if (!sony_row(a, grow, &s, &r))
    return;
  const LjStreamDev& S = a.streams[s];
  const int lane = threadIdx.x & 63;
  const uint32_t n = S.row_samples;
  const uint64_t row0 = uint64_t(r) * n;
  const int16_t* __restrict__ D = a.diffs + S.diff_offset + row0;
  const bool aligned = ((S.diff_offset + row0) & 7) == 0;
  int32_t sum = 0;
  for (uint32_t q = lane * 8; q < n; q += 512) {
    const uint4 t = sony_load8(D, q, n, aligned);
    sum += int16_t(t.x) + (int32_t(t.x) >> 16) + int16_t(t.y) + (int32_t(t.y) >> 16) +
           int16_t(t.z) + (int32_t(t.z) >> 16) + int16_t(t.w) + (int32_t(t.w) >> 16);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    sum += __shfl_xor(sum, o, 64);
  if (lane == 0)
    reinterpret_cast<int32_t*>(a.vseed)[uint64_t(grow) * 2 + 1] = sum;
}

__global__ __launch_bounds__(VS_T) void sony_vseed_kernel(LjArgs a) {
  __shared__ int32_t wtot[VS_T / 64];
  __shared__ int32_t carry_s;
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (S.kind != 2 || !a.nk[s].sony || a.results[s].status != 0)
    return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int32_t* __restrict__ V = reinterpret_cast<int32_t*>(a.vseed) + uint64_t(S.first_row) * 2;
  if (tid == 0)
    carry_s = 0; // "int pred = 0;" :67
  __syncthreads();
  for (uint32_t r0 = 0; r0 < S.rows; r0 += VS_T) {
    const uint32_t r = r0 + tid;
    const int32_t d = r < S.rows ? V[uint64_t(r) * 2 + 1] : 0;
    int32_t x = d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y = __shfl_up(x, o, 64);
      if (lane >= o)
        x += y;
    }
    if (lane == 63)
      wtot[wv] = x;
    __syncthreads();
    int32_t base = carry_s;
    for (int w = 0; w < wv; ++w)
      base += wtot[w];
    if (r < S.rows)
      V[uint64_t(r) * 2] = base + x - d; // exclusive
    __syncthreads();
    if (tid == VS_T - 1)
      carry_s = base + x;
    __syncthreads();
  }
}

// Values are written back over the differences (stream order, coalesced); the
// column scatter is sony_transpose_kernel's job.
__global__ __launch_bounds__(LJ_T) void sony_predict_kernel(LjArgs a) {
  const uint32_t grow = blockIdx.x * (LJ_T / 64) + (threadIdx.x >> 6);
  uint32_t s, r;
  if (!sony_row(a, grow, &s, &r))
    return;
  const LjStreamDev& S = a.streams[s];
  const int lane = threadIdx.x & 63;
  const uint32_t H = S.row_samples;
  const uint64_t row0 = uint64_t(r) * H;
  int16_t* __restrict__ D = a.diffs + S.diff_offset + row0;
  const bool aligned = ((S.diff_offset + row0) & 7) == 0;
  int32_t carry = reinterpret_cast<const int32_t*>(a.vseed)[uint64_t(grow) * 2];
  uint4 t0 = sony_load8(D, lane * 8, H, aligned);
  for (uint32_t q0 = 0; q0 < H; q0 += 512) {
    const uint32_t q = q0 + lane * 8;
    const uint4 t = t0;
    if (q0 + 512 < H)
      t0 = sony_load8(D, q + 512, H, aligned);
    int32_t v[8];
    v[0] = int16_t(t.x); v[1] = int32_t(t.x) >> 16;
    v[2] = int16_t(t.y); v[3] = int32_t(t.y) >> 16;
    v[4] = int16_t(t.z); v[5] = int32_t(t.z) >> 16;
    v[6] = int16_t(t.w); v[7] = int32_t(t.w) >> 16;
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      run += v[i];
      v[i] = run;
    }
    int32_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y = __shfl_up(x, o, 64);
      if (lane >= o)
        x += y;
    }
    const int32_t tot = __shfl(x, 63, 64);
    const int32_t e = x - run + carry;
    carry += tot;
    bool bad = false;
    uint32_t px[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int32_t p = v[i] + e;
      bad |= q + i < H && (uint32_t(p) >> 12) != 0; // !isIntN(pred, 12) :88
      px[i] = uint32_t(p) & 0xFFFFu;
    }
    if (bad)
      atomicCAS(&a.results[s].status, 0u, uint32_t(RSX_ERR_VALUE_RANGE));
    if (q >= H)
      continue;
    if (q + 8 <= H && aligned) {
      uint4 o;
      o.x = px[0] | (px[1] << 16);
      o.y = px[2] | (px[3] << 16);
      o.z = px[4] | (px[5] << 16);
      o.w = px[6] | (px[7] << 16);
      *reinterpret_cast<uint4*>(D + q) = o;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (q + i < H)
          D[q + i] = int16_t(px[i]);
    }
  }
}

// Stream order -> image: stream row r is image column W-1-r ("for (int col =
// out.width() - 1; col >= 0; col--)" :68), sample i of it is image row 2i for the
// first half and 2(i - H/2) + 1 for the second (:69-74).  64 x 64 tiles through
// LDS so that both the reads (along a stream row) and the writes (64 adjacent
// image columns) are contiguous.
constexpr int SONY_TILE = 64;
constexpr uint32_t SONY_TILES_R = (4600 + SONY_TILE - 1) / SONY_TILE; // w <= 4600 (.cpp:48)
constexpr uint32_t SONY_TILES_S = (3072 + SONY_TILE - 1) / SONY_TILE; // h <= 3072

__global__ __launch_bounds__(LJ_T) void sony_transpose_kernel(LjArgs a) {
  __shared__ uint16_t tile[SONY_TILE][SONY_TILE + 2];
  const uint32_t s = blockIdx.y;
  const LjStreamDev& S = a.streams[s];
  if (S.kind != 2 || !a.nk[s].sony || a.results[s].status != 0)
    return;
  const uint32_t W = S.rows, H = S.row_samples, half = H >> 1;
  const uint32_t r0 = (blockIdx.x / SONY_TILES_S) * SONY_TILE;
  const uint32_t i0 = (blockIdx.x % SONY_TILES_S) * SONY_TILE;
  if (r0 >= W || i0 >= H)
    return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint16_t* __restrict__ P =
      reinterpret_cast<const uint16_t*>(a.diffs + S.diff_offset);
#pragma unroll 4
  for (int k = 0; k < SONY_TILE / 4; ++k) {
    const uint32_t rr = r0 + wv * (SONY_TILE / 4) + k, ii = i0 + lane;
    tile[wv * (SONY_TILE / 4) + k][lane] =
        (rr < W && ii < H) ? P[uint64_t(rr) * H + ii] : uint16_t(0);
  }
  __syncthreads();
  uint8_t* img = a.out_base + S.img_offset;
  const uint32_t rr = r0 + lane;
#pragma unroll 4
  for (int k = 0; k < SONY_TILE / 4; ++k) {
    const uint32_t ii = i0 + wv * (SONY_TILE / 4) + k;
    if (rr < W && ii < H) {
      const uint32_t row = ii < half ? 2 * ii : 2 * (ii - half) + 1;
      reinterpret_cast<uint16_t*>(img + uint64_t(row) * S.img_pitch)[W - 1 - rr] =
          tile[lane][wv * (SONY_TILE / 4) + k];
    }
  }
}

} // namespace

void ljpeg_launch_reconstruct(const LjArgs& a, const ReconLaunch& r, hipStream_t s) {
  hipLaunchKernelGGL(lj_vseed_kernel, dim3(r.n_streams), dim3(VS_T), 0, s, a);
  if (r.any_nikon)
    hipLaunchKernelGGL(nk_vseed_kernel, dim3(r.n_streams), dim3(VS_T), 0, s, a);
  const dim3 grid((r.total_rows + 3) / 4), block(LJ_T);
  if (r.comp_present[1])
    hipLaunchKernelGGL((lj_predict_kernel<1>), grid, block, 0, s, a);
  if (r.comp_present[2])
    hipLaunchKernelGGL((lj_predict_fast_kernel<2>), grid, block, 0, s, a);
  if (r.comp_present[3])
    hipLaunchKernelGGL((lj_predict_kernel<3>), grid, block, 0, s, a);
  if (r.comp_present[4])
    hipLaunchKernelGGL((lj_predict_fast_kernel<4>), grid, block, 0, s, a);
  if (r.comp_present[5])
    hipLaunchKernelGGL((lj_predict_kernel<3, 4>), grid, block, 0, s, a);
  if (r.comp_present[6])
    hipLaunchKernelGGL((lj_predict_kernel<3, 6>), grid, block, 0, s, a);
  if (r.any_nikon)
    hipLaunchKernelGGL(nk_predict_kernel, grid, block, 0, s, a);
  if (r.any_sony) {
    hipLaunchKernelGGL(sony_rowsum_kernel, grid, block, 0, s, a);
    hipLaunchKernelGGL(sony_vseed_kernel, dim3(r.n_streams), dim3(VS_T), 0, s, a);
    hipLaunchKernelGGL(sony_predict_kernel, grid, block, 0, s, a);
    hipLaunchKernelGGL(sony_transpose_kernel, dim3(SONY_TILES_R * SONY_TILES_S, r.n_streams),
                       block, 0, s, a);
  }
}

} // namespace rsx
