// rsx_ljpeg_direct.hip -- fused decode + predictor reconstruction for LJPEG / CR2
// streams with 1, 2 or 4 interleaved components: the final decode writes pixels
// straight into the image; there is no difference scratch and no second pass.
//
// Replaces the inner loops of
//   LJpegDecompressor::decodeRowN  (decompressors/LJpegDecompressor.cpp:184-251, 326-332)
//   Cr2Decompressor::decompressN_X_Y  (decompressors/Cr2DecompressorImpl.h:431-465)
// for these shapes.  Arithmetic (model in tests/test_direct_recon_model.py;
// everything mod 2^16): with N components, row length RS and D the differences in
// stream order,
//   X[r][s] = (s >= N ? X[r][s-N] : (r ? X[r-1][s] : init[s])) + D[r][s].
// Let P(i) be the running sum over the WHOLE stream of the differences of i's
// component (no reset at row starts).  Then X(i) = P(i) + O(row(i), comp(i)) with
//   E(r, c) = P just before the row's first symbol,   F(r, c) = P at its c-th symbol,
//   V(r, c) = sum_{r' < r} (F(r', c) - E(r', c))        (the vertical chain),
//   O(r, c) = init[c] + V(r, c) - E(r, c).
// The synchronisation kernels (rsx_ljpeg.hip) have left the sums of every
// subsequence's differences (by relative phase) and lj_scan_kernel P before every
// workgroup, so:
//   K5a lj_rowedge  E and F of every stream row: the workgroups that hold a row
//                   start walk from the start of its subsequence to it
//   K5b lj_rowoff   O: one scan over the rows of each stream
//   K4d lj_decode_direct  final decode from validated start states; every lane turns
//                   the differences of its subsequence into pixels (running packed
//                   16-bit sums from its own P, plus O of the row) and stores them
//                   through the output mapping (tile crop, discarded MCUs, 2x2 MCUs,
//                   CR2 strips)
// Damaged streams (symbols past the end of the data) are flagged by lj_scan_kernel
// and take the legacy route instead.
#include "rsx_ljpeg_bits.h"

namespace rsx {

namespace {

// inclusive scan of x over the wavefront
__device__ __forceinline__ uint32_t wave_scan(uint32_t x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o)
      x += y;
  }
  return x;
}
__device__ __forceinline__ uint2 wave_scan_pk(uint2 x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint2 y = make_uint2(__shfl_up(x.x, o, 64), __shfl_up(x.y, o, 64));
    if (lane >= o)
      x = pk_add2(x, y);
  }
  return x;
}

// run-time flavour of lj_rot_fields (lanes of one wavefront may serve streams with
// different component counts)
__device__ __forceinline__ uint2 rot_fields_rt(uint2 v, uint32_t f, uint32_t n) {
  return n == 2 ? lj_rot_fields<2>(v, f) : (n == 4 ? lj_rot_fields<4>(v, f) : v);
}

// ---------------------------------------------------------------------------
// K5a: row edges, one lane per stream row.  The lane finds the workgroup and the
// subsequence its row's first symbol lies in (two binary searches over the scans the
// synchronisation kernels left: symbols before every workgroup / before every
// subsequence of it), copies that subsequence and the next one into its LDS column and
// walks from the subsequence's first symbol to the row start and on over the row's
// first MCU, adding up the differences on the way.  All rows of the launch walk at
// once (35 840 lanes for 8 cfg-3 frames): the first version had one walker per
// wavefront and workgroup of the entropy stream and spent 0.16 ms on ~60 serial steps
// of LDS latency each; this one takes as long as its longest walk.
// ---------------------------------------------------------------------------
constexpr int RE_T = 64;             // lanes per workgroup
constexpr int RE_SLOT_W = LJ_PW + 2; // 17 dwords of a subsequence + its data bits

struct WalkCursor {
  uint32_t b, slot, pos, ob, last_block;
};

// the staged copies (LDS, [dword][lane]): [0] = (b0, q0), [1] = the subsequence after it
struct WalkStage {
  const uint32_t* lds; // this lane's column: dword d of copy w at (w * RE_SLOT_W + d) * RE_T
  uint32_t b[2], q[2];
};

__device__ __forceinline__ const uint32_t* walk_image(const LjArgs& a, uint32_t b) {
  return reinterpret_cast<const uint32_t*>(a.unstuffed + size_t(b) * LJ_IMG_U4);
}

// move to the subsequence the position belongs to (a symbol belongs to the
// subsequence it starts in); false = ran off the end of the stream
__device__ __forceinline__ bool walk_normalise(WalkCursor& c, const LjArgs& a,
                                               const WalkStage& st) {
  while (c.pos >= c.ob) {
    c.pos -= c.ob;
    if (++c.slot == uint32_t(LJ_T)) {
      if (c.b == c.last_block)
        return false;
      ++c.b;
      c.slot = 1;
    }
    if (c.b == st.b[1] && c.slot == st.q[1])
      c.ob = st.lds[(RE_SLOT_W + LJ_PW + 1) * RE_T];
    else
      c.ob = walk_image(a, c.b)[LJ_BW * LJ_T + c.slot];
  }
  return true;
}

__device__ __forceinline__ uint32_t walk_window(const WalkCursor& c, const LjArgs& a,
                                                const WalkStage& st) {
  const uint32_t wi = c.pos >> 5;
  uint32_t d0, d1;
  if (c.b == st.b[0] && c.slot == st.q[0]) {
    d0 = st.lds[wi * RE_T];
    d1 = st.lds[(wi + 1) * RE_T];
  } else if (c.b == st.b[1] && c.slot == st.q[1]) {
    d0 = st.lds[(RE_SLOT_W + wi) * RE_T];
    d1 = st.lds[(RE_SLOT_W + wi + 1) * RE_T];
  } else {
    const uint32_t* B = walk_image(a, c.b);
    d0 = B[wi * LJ_T + c.slot];
    d1 = B[(wi + 1) * LJ_T + c.slot];
  }
  return uint32_t((((uint64_t(d0) << 32) | d1) << (c.pos & 31u)) >> 32);
}

__global__ __launch_bounds__(RE_T) void lj_rowedge_kernel(LjArgs a) {
  __shared__ uint32_t s_slots[2 * RE_SLOT_W * RE_T];
  const int lane = threadIdx.x;
  const uint32_t grow = blockIdx.x * RE_T + uint32_t(lane);
  if (grow >= a.total_rows)
    return;
  // row -> stream (first_row is increasing)
  uint32_t slo = 0, shi = a.n_streams - 1;
  while (slo < shi) {
    const uint32_t mid = (slo + shi + 1) >> 1;
    if (a.streams[mid].first_row <= grow)
      slo = mid;
    else
      shi = mid - 1;
  }
  const LjStreamDev& S = a.streams[slo];
  const uint32_t nd = S.direct;
  if (!nd || (a.results[slo].flags & FL_NEED_LEGACY) || !lj_pipeline_takes(a, slo, S))
    return;
  const uint32_t r = grow - S.first_row;
  const uint32_t RS = S.row_samples;
  const uint64_t t64 = uint64_t(r) * RS;
  if (r >= S.rows || t64 >= S.needed)
    return;
  const uint32_t t = uint32_t(t64); // the row's first symbol
  // the last workgroup with base <= t
  const uint32_t fb = S.first_block;
  uint32_t lo = 0, hi = S.n_blocks - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (a.block_base[fb + mid] <= t)
      lo = mid;
    else
      hi = mid - 1;
  }
  const uint32_t b = fb + lo;
  const uint32_t base = a.block_base[b];
  const uint32_t tl = t - base; // workgroup-relative index of the row's first symbol
  if (tl >= a.block_sum[b])
    return; // past the end of the data: the stream is flagged for the legacy route
  // the last slot of it whose first symbol is at or before tl
  const uint32_t g0 = S.first_subseq + lo * LJ_OWN; // record of slot 1
  uint32_t qlo = 0, qhi = LJ_OWN - 1;
  while (qlo < qhi) {
    const uint32_t mid = (qlo + qhi + 1) >> 1;
    if (a.sub_first[g0 + mid] <= tl)
      qlo = mid;
    else
      qhi = mid - 1;
  }
  const uint32_t q = qlo + 1;
  uint32_t skip = tl - a.sub_first[g0 + qlo];
  // P before the slot's first symbol, by absolute component
  uint2 P = pk_add2(a.block_pbase[b], rot_fields_rt(a.sub_psum[g0 + qlo], base % nd, nd));
  const uint32_t st0 = q == 1 ? a.block_start[b] : (a.sub_state[g0 + qlo - 1] & ST_MASK);
  const uint32_t last_block = fb + S.n_blocks - 1;
  // stage the slot and its successor
  WalkStage st;
  st.lds = s_slots + lane;
  st.b[0] = b;
  st.q[0] = q;
  st.b[1] = q + 1 < uint32_t(LJ_T) ? b : b + 1;
  st.q[1] = q + 1 < uint32_t(LJ_T) ? q + 1 : 1u;
  const bool have_next = st.b[1] <= last_block;
  {
    const uint32_t* B0 = walk_image(a, b);
    const uint32_t* B1 = walk_image(a, have_next ? st.b[1] : b);
    const uint32_t q1 = have_next ? st.q[1] : q;
#pragma unroll
    for (int d = 0; d <= LJ_PW; ++d) {
      s_slots[d * RE_T + lane] = B0[uint32_t(d) * LJ_T + q];
      s_slots[(RE_SLOT_W + d) * RE_T + lane] = B1[uint32_t(d) * LJ_T + q1];
    }
    s_slots[(LJ_PW + 1) * RE_T + lane] = B0[LJ_BW * LJ_T + q];
    s_slots[(RE_SLOT_W + LJ_PW + 1) * RE_T + lane] = B1[LJ_BW * LJ_T + q1];
  }
  if (!have_next)
    st.b[1] = 0xFFFFFFFFu;
  uint32_t idx = base + a.sub_first[g0 + qlo]; // absolute index of the next symbol
  WalkCursor c;
  c.b = b;
  c.slot = q;
  c.pos = st0 & ST_OFF_MASK;
  c.ob = st.lds[(LJ_PW + 1) * RE_T];
  c.last_block = last_block;
  bool ok = !(st0 & ST_ERR);
  uint2 E = make_uint2(0, 0);
  const TabLds* tabs = a.tables + S.table_base;
  const bool multi = S.n_tables > 1;
  if (LJ_ABLATE & 64u)
    skip = 0;
  // `skip` symbols up to the row start, then the nd symbols of its first MCU
  for (uint32_t n = 0; ok && n < skip + nd; ++n) {
    if (n == skip)
      E = P;
    if (!walk_normalise(c, a, st)) {
      ok = false;
      break;
    }
    const uint32_t ph = idx % nd;
    const uint32_t w = walk_window(c, a, st);
    const uint32_t e = lj_entry_global(w, tabs + (multi ? S.tab_of_phase[ph] : 0));
    if (e == 0u) {
      ok = false;
      break;
    }
    const uint32_t d = lj_extend(w, e);
    const uint32_t v = d << (16u * (ph & 1u));
    P = (ph & 2u) ? make_uint2(P.x, pk_add(P.y, v)) : make_uint2(pk_add(P.x, v), P.y);
    c.pos += e >> 10;
    ++idx;
  }
  // (a stream in error never shows its pixels: the values do not matter then)
  a.row_edge[uint64_t(S.first_row) + r] = make_uint4(E.x, E.y, P.x, P.y);
}

// ---------------------------------------------------------------------------
// K5b: row offsets O(r, c) = init[c] + V(r, c) - E(r, c), V = exclusive scan over
// the rows of F - E.  One workgroup per stream.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(VS_T) void lj_rowoff_kernel(LjArgs a) {
  __shared__ uint2 wtot[VS_T / 64];
  __shared__ uint2 carry_s;
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (!S.direct || (a.results[s].flags & FL_NEED_LEGACY) || !lj_pipeline_takes(a, s, S))
    return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // only the rows that are decoded have edges
  const uint64_t last_sym = S.needed ? S.needed - 1 : 0;
  const uint32_t rows = S.needed ? uint32_t(last_sym / S.row_samples) + 1 : 0u;
  const uint2 init = make_uint2(uint32_t(S.init_pred[0]) | (uint32_t(S.init_pred[1]) << 16),
                                uint32_t(S.init_pred[2]) | (uint32_t(S.init_pred[3]) << 16));
  const uint4* __restrict__ RE = a.row_edge + S.first_row;
  uint2* __restrict__ O = reinterpret_cast<uint2*>(a.vseed) + S.first_row;
  if (tid == 0)
    carry_s = make_uint2(0, 0);
  __syncthreads();
  for (uint32_t rb = 0; rb < rows; rb += VS_T) {
    const uint32_t r = rb + tid;
    uint4 ef = make_uint4(0, 0, 0, 0);
    if (r < rows)
      ef = RE[r];
    const uint2 E = make_uint2(ef.x, ef.y), F = make_uint2(ef.z, ef.w);
    const uint2 d = pk_sub2(F, E);
    const uint2 incl = wave_scan_pk(d, lane);
    if (lane == 63)
      wtot[wv] = incl;
    __syncthreads();
    uint2 v = pk_add2(carry_s, pk_sub2(incl, d));
    for (int w = 0; w < wv; ++w)
      v = pk_add2(v, wtot[w]);
    if (r < rows)
      O[r] = pk_sub2(pk_add2(init, v), E);
    __syncthreads();
    if (tid == VS_T - 1)
      carry_s = pk_add2(v, d);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// K4d: final decode straight into the image
// ---------------------------------------------------------------------------
// where the next sample of a lane goes: the current run of samples that are stored
// contiguously (or all discarded) and the offsets of its row
struct OutCursor {
  uint16_t* p;   // address of the next sample (nullptr: the run is discarded)
  uint32_t left; // samples left in the run (0: no contiguous run -- per-sample path)
  uint2 off;     // O(row, .) by the lane's relative phases
};

// what the store paths need to know
struct StoreCtx {
  uint8_t* img;          // out_base + img_offset
  const uint2* rowoff;   // O of the stream's rows
  const Cr2Strip* strips;
  const LjStreamDev* S;
};

__device__ __forceinline__ void strip_divmod32(uint64_t off, uint32_t w, uint32_t* row,
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

// one sample through the general output mapping (decodeRowN :200-250 / CR2 strips)
__device__ __forceinline__ void lj_put_sample(const StoreCtx& x, uint32_t r, uint32_t sidx,
                                              uint16_t val) {
  const LjStreamDev& S = *x.S;
  if (S.kind == 0) {
    const uint32_t m = sidx / S.n_comp, c = sidx - m * S.n_comp;
    const uint32_t col = S.mcu_w * m + (c % S.mcu_w);
    if (col >= S.keep_samples)
      return;
    const uint32_t row = S.out_y + S.mcu_h * r + c / S.mcu_w;
    reinterpret_cast<uint16_t*>(x.img + uint64_t(row) * S.img_pitch)[S.out_x + col] = val;
  } else {
    const uint64_t k = uint64_t(r) * S.row_samples + sidx;
    const Cr2Strip* st = x.strips;
    uint32_t q = 0;
    while (q + 1 < S.n_strips && k >= st[q + 1].first_sample)
      ++q;
    uint32_t srow, scol;
    strip_divmod32(k - st[q].first_sample, st[q].w, &srow, &scol);
    reinterpret_cast<uint16_t*>(x.img + uint64_t(st[q].y0 + srow) * S.img_pitch)[st[q].x0 + scol] =
        val;
  }
}

// the run that sample i of the stream starts (once per lane, and again at the end of
// every run)
template <int N>
__device__ __forceinline__ OutCursor lj_locate(const StoreCtx& x, uint32_t i, uint32_t rot) {
  const LjStreamDev& S = *x.S;
  OutCursor c;
  const uint32_t RS = S.row_samples;
  const uint32_t r = i / RS, sidx = i - r * RS;
  c.off = lj_rot_fields<N>(x.rowoff[r], rot);
  c.p = nullptr;
  c.left = 0;
  if (S.kind == 0) {
    if (S.mcu_h == 1) { // sample s of a stream row = output column s
      if (sidx < S.keep_samples) {
        c.left = (S.keep_samples < RS ? S.keep_samples : RS) - sidx;
        c.p = reinterpret_cast<uint16_t*>(x.img + uint64_t(S.out_y + r) * S.img_pitch) +
              S.out_x + sidx;
      } else {
        c.left = RS - sidx; // trailing MCUs of the frame that the tile does not keep
      }
    }
  } else {
    const Cr2Strip* st = x.strips;
    uint32_t z = 0;
    while (z + 1 < S.n_strips && uint64_t(i) >= st[z + 1].first_sample)
      ++z;
    uint32_t srow, col;
    strip_divmod32(uint64_t(i) - st[z].first_sample, st[z].w, &srow, &col);
    const uint32_t in_strip = st[z].w - col, in_row = RS - sidx;
    c.left = in_strip < in_row ? in_strip : in_row;
    c.p = reinterpret_cast<uint16_t*>(x.img + uint64_t(st[z].y0 + srow) * S.img_pitch) +
          st[z].x0 + col;
  }
  return c;
}

// Everything that is not "a full group of 8 inside a stored run": groups that cross the
// end of a run (row end, kept width, strip), discarded runs, mappings without
// contiguous runs (2x2 MCUs), and -- when it is not inside one run -- the lane's last,
// partial group.  pv = the group's running sums P (sample t in half t&1 of dword
// t>>1), i = index of its first sample.  The group is cut at the run ends: each piece
// is stored (or dropped) as a whole and the cursor moved on, so the row offsets are
// loaded once per run, not per sample.  Returns the cursor for sample i + cnt.
template <int N>
__device__ __forceinline__ OutCursor lj_store_general(const StoreCtx& x, OutCursor oc,
                                                      uint32_t i, uint32_t rot, uint4 pv,
                                                      uint32_t cnt) {
  auto pick = [&](uint32_t t) -> uint32_t { // sample t of the group
    const uint32_t k = t >> 1;
    const uint32_t d = k == 0 ? pv.x : (k == 1 ? pv.y : (k == 2 ? pv.z : pv.w));
    return (d >> (16u * (t & 1u))) & 0xFFFFu;
  };
  uint32_t t = 0;
#pragma unroll 1
  while (t < cnt) {
    if (oc.left == 0)
      oc = lj_locate<N>(x, i + t, rot);
    uint32_t n = cnt - t < oc.left ? cnt - t : oc.left;
    if (n == 0) {
      // no contiguous runs in this mapping: sample by sample
      const uint32_t RS = x.S->row_samples;
      uint32_t rr = (i + t) / RS, ss = (i + t) - rr * RS;
#pragma unroll 1
      for (; t < cnt; ++t) {
        const uint32_t c = (i + t) & uint32_t(N - 1);
        const uint2 o = x.rowoff[rr];
        const uint32_t ov = ((c & 2u) ? o.y : o.x) >> (16u * (c & 1u));
        lj_put_sample(x, rr, ss, uint16_t(pick(t) + ov));
        if (++ss == RS) {
          ss = 0;
          ++rr;
        }
      }
      break;
    }
    if (oc.p) {
      const uint32_t o0 = N == 1 ? (oc.off.x & 0xFFFFu) * 0x10001u : oc.off.x;
      const uint32_t o1 = N == 4 ? oc.off.y : o0;
#pragma unroll 1
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t q = t + k; // (the offsets go by the sample's phase: q mod N)
        const uint32_t o = (q & 2u) ? o1 : o0;
        oc.p[k] = uint16_t(pick(q) + (o >> (16u * (q & 1u))));
      }
      oc.p += n;
    }
    oc.left -= n;
    t += n;
  }
  return oc;
}

#ifndef RSX_K4D_BURST
#define RSX_K4D_BURST 4
#endif
constexpr int K4D_BURST = RSX_K4D_BURST; // groups of 8 samples decoded before their stores

#ifndef RSX_K4D_LDS_PAD
#define RSX_K4D_LDS_PAD 0 // (experiments: unused LDS that limits the workgroups per CU)
#endif
constexpr size_t k4d_lds_bytes(int n_tables) {
  return size_t(LJ_BW_DEC) * LJ_T * 4 + 16 * 4 + size_t(n_tables) * sizeof(TabLds) +
         RSX_K4D_LDS_PAD;
}

#ifndef RSX_K4D_MIN_WAVES
#define RSX_K4D_MIN_WAVES 1
#endif
template <bool MULTI, int N>
__global__ __launch_bounds__(LJ_T, RSX_K4D_MIN_WAVES) void lj_decode_direct_kernel(LjArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x;
  const uint32_t s = a.block_stream[b];
  const LjStreamDev& S = a.streams[s];
  if ((S.n_tables > 1) != MULTI || int(S.direct) != N)
    return;
  if ((a.results[s].flags & FL_NEED_LEGACY) || !lj_pipeline_takes(a, s, S))
    return;
  Lds L{};
  // (the tables first: a single table's LUT then starts at LDS address 0 and its entries
  // are addressed with an OR; the image lies with its dword rows reversed, see lj_window)
  L.tabs = reinterpret_cast<TabLds*>(smem);
  L.B = reinterpret_cast<uint32_t*>(smem + size_t(MULTI ? S.n_tables : 1u) * sizeof(TabLds));
  L.misc = L.B + LJ_BW_DEC * LJ_T;
  const uint32_t lb = b - S.first_block;
  const int j = threadIdx.x;
  const uint64_t needed = S.needed;
  const uint32_t base = a.block_base[b];
  const uint32_t sum = a.block_sum[b];
  if (base >= needed || sum == 0)
    return; // nothing of this workgroup is delivered
  uint64_t M = a.results[s].marker_pos;
  if (M > lj_data_end(S))
    M = lj_data_end(S);
  if (uint64_t(lb) * LJ_R > M)
    return; // past the end of data

  lj_stage_tables(L, a, S);
  // the image (no ob[] here: the records say how many symbols a slot holds): all
  // loads are issued before the first LDS store (written as one loop the compiler
  // waits for every load before it issues the next)
  {
    const uint4* __restrict__ src = a.unstuffed + size_t(b) * LJ_IMG_U4;
    uint4* dst = reinterpret_cast<uint4*>(L.B);
    constexpr int n4 = LJ_BW_DEC * LJ_T / 4, n_it = (n4 + LJ_T - 1) / LJ_T;
    uint4 t[n_it];
#pragma unroll
    for (int h = 0; h < n_it; ++h) {
      const int i = h * LJ_T + j;
      t[h] = make_uint4(0, 0, 0, 0);
      if (i < n4) {
#ifdef RSX_K4D_NT_LOADS
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + i));
        t[h] = make_uint4(q.x, q.y, q.z, q.w);
#else
        t[h] = src[i];
#endif
      }
    }
#pragma unroll
    for (int h = 0; h < n_it; ++h) {
      const int i = h * LJ_T + j;
      if (i < n4) // uint4 i = dwords 4 * (i % 64) .. of row i / 64 -> row BW - 1 - i / 64
        dst[(LJ_BW_DEC - 1 - (i >> 6)) * (LJ_T / 4) + (i & 63)] = t[h];
    }
  }
  DecodeParams dp = lj_params(S);
  const uint32_t g0 = S.first_subseq + lb * LJ_OWN;
  const uint32_t gsub = g0 + uint32_t(j - 1);
  // the lane's records: symbols and running sums P before its slot (both inside the
  // workgroup; left by the synchronisation kernels)
  uint32_t my_rec = 0, before = 0;
  uint2 pex = make_uint2(0, 0);
  if (j >= 1) {
    my_rec = a.sub_state[gsub];
    before = a.sub_first[gsub];
    pex = a.sub_psum[gsub];
  }
  __syncthreads(); // image + tables complete
  dp.long_codes = lj_long_codes(L, S.n_tables);
  const uint32_t my_count = my_rec >> 16, my_exit = my_rec & ST_MASK;
  uint32_t my_start = 0;
  if (j >= 1)
    my_start = (j == 1) ? a.block_start[b] : (a.sub_state[gsub - 1] & ST_MASK);
  const uint64_t first = uint64_t(base) + before; // first symbol of this lane

  // a bad Huffman code inside the delivered range is a real error
  // (PrefixCodeLookupDecoder.h:152-155)
  if (j >= 1 && (my_exit & ST_ERR) && first + my_count < needed &&
      int64_t(lb) * LJ_R + int64_t(j - 1) * LJ_P < int64_t(M))
    atomicCAS(&a.results[s].status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));

  uint32_t remaining = (my_start & ST_ERR) ? 0u : my_count;
  if (first >= needed)
    remaining = 0;
  else if (first + remaining > needed)
    remaining = uint32_t(needed - first);

  uint32_t wmax = remaining;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    wmax = max(wmax, uint32_t(__shfl_xor(wmax, o, 64)));
  const uint32_t n_groups = (LJ_ABLATE & 2u) ? 0u : (wmax + 7) >> 3;

  // P before the lane's first symbol by absolute component, then by the lane's own
  // relative phases (the k-th symbol of the lane has phase k mod N: groups of 8
  // keep that alignment)
  const uint32_t i0 = uint32_t(first);
  const uint32_t rot = (0u - i0) & uint32_t(N - 1);
  const uint2 p_abs =
      pk_add2(a.block_pbase[b], lj_rot_fields<N>(pex, base & uint32_t(N - 1)));
  uint2 run = lj_rot_fields<N>(p_abs, rot);
  if (N == 1)
    run.x = (run.x & 0xFFFFu) * 0x10001u; // both halves carry the one running sum
  StoreCtx sx;
  sx.img = a.out_base + S.img_offset;
  sx.rowoff = reinterpret_cast<const uint2*>(a.vseed) + S.first_row;
  sx.strips = a.strips + S.strip_base;
  sx.S = &S;
  OutCursor oc{};
  if (LJ_ABLATE & 16384u) {
    oc.p = reinterpret_cast<uint16_t*>(sx.img) + i0;
    oc.left = 0x7FFFFFFFu;
  } else if (remaining) {
    oc = lj_locate<N>(sx, i0, rot);
  }
  uint32_t phase = (my_start >> ST_PHASE_SHIFT) & 7u;
  // Window reader on integer LDS addresses (this kernel is bound by VALU issue, too: the
  // register bit reader it had keeps the window load off the critical path but costs 11
  // instructions more per symbol).
  uint32_t pos = my_start & ST_OFF_MASK;
  const uint32_t vrow = lds_addr(&L.B[(LJ_BW_DEC - 2) * LJ_T + j]); // row of dword 1
  const uint32_t lut0 = lds_addr(L.tabs[0].lut);
  auto decode_group = [&](uint32_t g, uint32_t (&p)[4]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool live = 8 * g + q < remaining;
      const uint32_t ad = vrow + uint32_t(__mul24(int(pos >> 5), -4 * LJ_T));
      const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
      const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << (pos & 31u)) >> 32);
      uint32_t d, e;
      if (MULTI || RSX_LUT_DIFF) {
        e = lj_entry_diff(w, lj_table<MULTI>(L, dp, phase), live, dp.long_codes, &d);
      } else {
        e = *(lds_u16p)(lut0 | ((w >> (31 - LUT_BITS)) & ((2u << LUT_BITS) - 2u)));
        if (dp.long_codes && __builtin_expect(__any(live && (e & 31u) == 0u), 0)) {
          if (live && (e & 31u) == 0u)
            e = lj_slow_entry(w, &L.tabs[0]);
        }
        d = lj_extend(w, e);
      }
      pos += live ? (e >> 10) : 0u;
      if (MULTI)
        phase = live ? ((phase + 1 == dp.period) ? 0u : phase + 1) : phase;
      const uint32_t diff = live ? d : 0u;
      if (q & 1)
        p[q >> 1] |= diff << 16;
      else
        p[q >> 1] = diff;
    }
    // differences -> running sums P (packed; dword k holds samples 2k, 2k+1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (N == 1) {
        // (d0, d1) -> (run + d0, run + d0 + d1)
        const uint32_t t = pk_add(p[k], p[k] << 16);
        p[k] = pk_add(t, run.x);
        run.x = (p[k] >> 16) * 0x10001u;
      } else if (N == 2) {
        run.x = pk_add(run.x, p[k]);
        p[k] = run.x;
      } else if (k & 1) {
        run.y = pk_add(run.y, p[k]);
        p[k] = run.y;
      } else {
        run.x = pk_add(run.x, p[k]);
        p[k] = run.x;
      }
    }
  };
  // the packed offsets of the current row for dwords 0/2 and 1/3 of a group
  auto off0 = [&]() { return N == 1 ? (oc.off.x & 0xFFFFu) * 0x10001u : oc.off.x; };
  auto off1 = [&]() { return N == 4 ? oc.off.y : off0(); };
  uint32_t tp[4] = {0, 0, 0, 0}; // the lane's last, partial group
  for (uint32_t gb = 0; gb < n_groups; gb += K4D_BURST) {
    uint32_t pv[K4D_BURST][4];
#pragma unroll
    for (int u = 0; u < K4D_BURST; ++u)
      if (gb + u < n_groups) // wave-uniform
        decode_group(gb + u, pv[u]);
    if (LJ_ABLATE & 1u) {
#pragma unroll
      for (int u = 0; u < K4D_BURST; ++u)
        asm volatile("" ::"v"(pv[u][0]), "v"(pv[u][1]), "v"(pv[u][2]), "v"(pv[u][3]));
      continue;
    }
    // Full groups inside a stored run leave as one (unaligned) 16-byte store each.
    // A lane whose group ends a run (or is discarded) stops here: that group and the
    // ones after it in the burst are marked and go through the general path below, in
    // order (the general path moves the cursor).
    uint32_t pend = 0;
#pragma unroll
    for (int u = 0; u < K4D_BURST; ++u) {
      const uint32_t g = gb + uint32_t(u);
      if (g >= n_groups)
        break;
      const bool full = 8 * g + 8 <= remaining;
      if (!full && 8 * g < remaining) {
        tp[0] = pv[u][0];
        tp[1] = pv[u][1];
        tp[2] = pv[u][2];
        tp[3] = pv[u][3];
      }
      if (full && pend == 0 && oc.left >= 8 && oc.p != nullptr) {
        const uint32_t o0 = off0(), o1 = off1();
        const uint4 v = make_uint4(pk_add(pv[u][0], o0), pk_add(pv[u][1], o1),
                                   pk_add(pv[u][2], o0), pk_add(pv[u][3], o1));
        if (LJ_ABLATE & 1024u)
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(oc.p));
        else
          __builtin_memcpy(oc.p, &v, 16);
        oc.p += 8;
        oc.left -= 8;
      } else if (full) {
        pend |= 1u << u;
      }
    }
    if (!(LJ_ABLATE & 4096u) && __any(pend != 0)) {
#pragma unroll 1
      for (uint32_t u = 0; u < uint32_t(K4D_BURST); ++u) {
        uint32_t p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t v = pv[0][k];
#pragma unroll
          for (int q = 1; q < K4D_BURST; ++q)
            v = u == uint32_t(q) ? pv[q][k] : v;
          p[k] = v;
        }
        if ((pend >> u) & 1u)
          oc = lj_store_general<N>(sx, oc, i0 + 8 * (gb + u), rot,
                                   make_uint4(p[0], p[1], p[2], p[3]), 8u);
      }
    }
  }
  // the partial last group: inside one stored run (the rule) it leaves as at most
  // three stores -- 4, 2 and 1 samples -- instead of up to seven 2-byte ones (a
  // scattered store instruction costs the memory pipeline the same whatever its width)
  if (!(LJ_ABLATE & (1u | 8192u)) && (remaining & 7u) != 0) {
    const uint32_t cnt = remaining & 7u;
    if (cnt <= oc.left && oc.p != nullptr) {
      const uint32_t o0 = off0(), o1 = off1();
      uint32_t w0 = pk_add(tp[0], o0), w1 = pk_add(tp[1], o1), w2 = pk_add(tp[2], o0),
               w3 = pk_add(tp[3], o1);
      uint16_t* q = oc.p;
      if (cnt & 4u) {
        const uint2 v = make_uint2(w0, w1);
        __builtin_memcpy(q, &v, 8);
        q += 4;
        w0 = w2;
        w1 = w3;
      }
      if (cnt & 2u) {
        __builtin_memcpy(q, &w0, 4);
        q += 2;
        w0 = w1;
      }
      if (cnt & 1u)
        *q = uint16_t(w0);
    } else {
      (void)lj_store_general<N>(sx, oc, i0 + (remaining & ~7u), rot,
                                make_uint4(tp[0], tp[1], tp[2], tp[3]), cnt);
    }
  }
  // K7 needs the bit position at which the reference's last symbol starts:
  // exactly one lane of the whole stream owns it and walks there again
  if (needed >= 1 && needed - 1 >= first && needed - 1 < first + remaining) {
    const uint32_t target = uint32_t(needed - 1 - first);
    uint32_t p2 = my_start & ST_OFF_MASK, ph2 = (my_start >> ST_PHASE_SHIFT) & 7u;
    for (uint32_t t = 0; t < target; ++t) {
      const uint32_t w = lj_window<LJ_BW_DEC>(L.B, j, p2);
      const TabLds& tb = lj_table<MULTI>(L, dp, ph2);
      uint32_t e = lj_lut16(tb, w >> (32 - LUT_BITS));
      if ((e & 31u) == 0u)
        e = lj_slow_entry(w, &tb);
      p2 += e >> 10;
      if (MULTI)
        ph2 = (ph2 + 1 == dp.period) ? 0u : ph2 + 1;
    }
    a.results[s].last_slot = lb * LJ_OWN + uint32_t(j - 1);
    a.results[s].last_pos = p2;
  }
}

template <bool MULTI, int N>
void launch_direct_one(const LjArgs& a, const DirectLaunch& d, hipStream_t s,
                       KernelTimer* timer) {
  if (!d.present[MULTI ? 1 : 0][N])
    return;
  hipLaunchKernelGGL((lj_decode_direct_kernel<MULTI, N>), dim3(d.total_blocks), dim3(LJ_T),
                     k4d_lds_bytes(MULTI ? d.max_tables : 1), s, a);
  if (timer)
    timer->mark(MULTI ? "lj_decode_direct_kernel<multi>" : "lj_decode_direct_kernel");
}

} // namespace

void ljpeg_launch_direct(const LjArgs& a, const DirectLaunch& d, hipStream_t s,
                         KernelTimer* timer) {
  const bool n1 = d.present[0][1] || d.present[1][1];
  const bool n2 = d.present[0][2] || d.present[1][2];
  const bool n4 = d.present[0][4] || d.present[1][4];
  (void)n1;
  (void)n2;
  (void)n4;
  hipLaunchKernelGGL(lj_rowedge_kernel, dim3((d.total_rows + RE_T - 1) / RE_T), dim3(RE_T), 0,
                     s, a);
  if (timer)
    timer->mark("lj_rowedge_kernel");
  hipLaunchKernelGGL(lj_rowoff_kernel, dim3(d.n_streams), dim3(VS_T), 0, s, a);
  if (timer)
    timer->mark("lj_rowoff_kernel");
  launch_direct_one<false, 1>(a, d, s, timer);
  launch_direct_one<false, 2>(a, d, s, timer);
  launch_direct_one<false, 4>(a, d, s, timer);
  launch_direct_one<true, 1>(a, d, s, timer);
  launch_direct_one<true, 2>(a, d, s, timer);
  launch_direct_one<true, 4>(a, d, s, timer);
}

} // namespace rsx
