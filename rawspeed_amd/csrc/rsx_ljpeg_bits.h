// rsx_ljpeg_bits.h -- device-side building blocks shared by the kernels of the
// lossless-JPEG family pipeline (rsx_ljpeg.hip: un-stuffing, synchronisation,
// legacy decode; rsx_ljpeg_direct.hip: the fused decode + reconstruction):
// the per-workgroup LDS image of un-stuffed subsequences, the two bit readers,
// the code-table lookup and the JPEG "EXTEND" of a difference.
//
// Symbol semantics: codes/AbstractPrefixCodeDecoder.h:43-76,
// codes/PrefixCodeLUTDecoder.h:150-216.
#pragma once

#include "rsx_ljpeg_dev.h"

namespace rsx {

// LDS image of a workgroup.  B is [dword][slot] (dword k of slot j at k*LJ_T+j:
// lane j always hits bank j%32); the 16-bit records follow.
struct Lds {
  uint32_t* B;    // [bw][LJ_T] big-endian dwords of every slot, un-stuffed
  uint16_t* su;   // [LJ_T] start state each slot was last decoded from
  uint16_t* st;   // [LJ_T] exit state of each slot
  uint16_t* cn;   // [LJ_T] symbols that start inside each slot (<= 512)
  uint16_t* ob;   // [LJ_T] data bits of each slot's own 64 bytes (<= 512)
  uint16_t* list; // [LJ_T] dense list of slots to re-decode
  uint32_t* sm;   // [2*LJ_T] per slot: sums of its differences by relative phase
                  //          (4 x u16; K0 parks its 16 byte-compaction selectors here)
  uint32_t* misc; // [16]
  TabLds* tabs;
  uint32_t* rec;  // [LJ_T] synchronisation kernels only: su, st, cn packed (carve_sync)
};

// Sized to the byte: gfx950 hands out LDS in 1280-byte granules (160 KB / 128) and
// the synchronisation kernels are latency bound -- their speed is the number of
// resident workgroups.  So the records are 16-bit and the kernels keep only the
// dwords of a slot they can touch: the un-stuffer all LJ_BW = 20, the window reader
// 17 -- a live symbol starts before bit 512, its 32-bit window ends in dword 16 --
// or 18 for pair symbols (second code <= 16 bits later) and for the register bit
// reader (which prefetches one dword past its 64-bit buffer).
constexpr int LJ_BW_SYNC = LJ_PW + 1, LJ_BW_SYNC_PAIR = LJ_PW + 2, LJ_BW_DEC = LJ_PW + 2;
constexpr size_t lj_lds_words(int bw) {
  return size_t(bw) * LJ_T + 4 * (LJ_T / 2) + 2 * LJ_T + LJ_T / 2 + 16;
}

__device__ __forceinline__ Lds carve(uint8_t* smem, int bw = LJ_BW) {
  Lds l;
  l.B = reinterpret_cast<uint32_t*>(smem);
  l.su = reinterpret_cast<uint16_t*>(l.B + bw * LJ_T);
  l.st = l.su + LJ_T;
  l.cn = l.st + LJ_T;
  l.ob = l.cn + LJ_T;
  l.sm = reinterpret_cast<uint32_t*>(l.ob + LJ_T);
  l.list = reinterpret_cast<uint16_t*>(l.sm + 2 * LJ_T);
  l.misc = l.sm + 2 * LJ_T + LJ_T / 2;
  l.tabs = reinterpret_cast<TabLds*>(l.misc + 16);
  return l;
}

constexpr size_t lj_lds_bytes(int n_tables, int bw = LJ_BW) {
  return lj_lds_words(bw) * 4 + size_t(n_tables) * sizeof(TabLds);
}

// The synchronisation kernels' own, tighter layout.  Their speed is the number of
// workgroups a CU holds (measured with padded allocations, 8 cfg-3 frames: 3 / 4 / 5 / 6
// workgroups = 0.785 / 0.617 / 0.525 / 0.472 ms), so everything is packed to get a
// SEVENTH one in (18 granules of 1280 bytes):
//  * one 32-bit record per slot instead of three 16-bit arrays: exit state (10 bits:
//    ST_ERR | phase | offset), the start state it was decoded from (10), symbols (12);
//  * the difference sums take one dword per slot for up to two components, none when
//    the stream does not take the fused path;
//  * single-table streams look their codes up on 10 bits instead of 11 (TabLds10: 2 KB
//    less; codes of 11 bits and more -- the rarest categories -- take the search).
struct alignas(16) TabLds10 {
  uint16_t lut[1024];
  uint32_t max_code[18];
  uint16_t val_offset[18];
  uint8_t values[RSX_MAX_CODE_VALUES];
  uint8_t max_len;
  uint8_t fix16;
  uint8_t zero_sym_bits;
  uint8_t las;
};
static_assert(sizeof(TabLds10) % 16 == 0, "TabLds10 must be 16-byte sized");
template <typename TB> struct TabBits { static constexpr int value = LUT_BITS; };
template <> struct TabBits<TabLds10> { static constexpr int value = 10; };

constexpr uint32_t REC_ST_MASK = 0x3FFu, REC_SU_SHIFT = 10, REC_CN_SHIFT = 20;
static_assert(ST_ERR < 1024u, "a state fits the record's 10 bits");
__device__ __forceinline__ uint32_t rec_make(uint32_t su, uint32_t st, uint32_t cn) {
  return (st & REC_ST_MASK) | ((su & REC_ST_MASK) << REC_SU_SHIFT) | (cn << REC_CN_SHIFT);
}
__device__ __forceinline__ uint32_t rec_st(uint32_t r) { return r & REC_ST_MASK; }
__device__ __forceinline__ uint32_t rec_su(uint32_t r) { return (r >> REC_SU_SHIFT) & REC_ST_MASK; }
__device__ __forceinline__ uint32_t rec_cn(uint32_t r) { return r >> REC_CN_SHIFT; }

constexpr size_t lj_sync_sm_words(int ns) { return ns == 0 ? 0 : (ns <= 2 ? LJ_T : 2 * LJ_T); }
constexpr size_t lj_sync_lds_words(int bw, int ns) {
  return size_t(bw) * LJ_T + LJ_T + LJ_T / 2 + lj_sync_sm_words(ns) + LJ_T / 2 + 16;
}
template <typename TB>
constexpr size_t lj_sync_lds_bytes(int n_tables, int bw, int ns) {
  return lj_sync_lds_words(bw, ns) * 4 + size_t(n_tables) * sizeof(TB);
}
static_assert(lj_sync_lds_bytes<TabLds10>(1, LJ_BW_SYNC, 2) <= 18 * 1280,
              "seven synchronisation workgroups per CU");
static_assert(lj_sync_lds_words(LJ_BW_SYNC, 0) % 4 == 0 && lj_sync_lds_words(LJ_BW_SYNC_PAIR, 4) % 4 == 0,
              "the tables start on a 16-byte boundary");

// (the tables come FIRST: the single table's LUT then starts at LDS address 0, and the
// lookup forms its address with an OR instead of an add)
__device__ __forceinline__ Lds carve_sync(uint8_t* smem, int bw, int ns, size_t table_bytes) {
  Lds l{};
  l.tabs = reinterpret_cast<TabLds*>(smem);
  l.B = reinterpret_cast<uint32_t*>(smem + table_bytes);
  l.rec = l.B + bw * LJ_T;
  l.ob = reinterpret_cast<uint16_t*>(l.rec + LJ_T);
  l.sm = reinterpret_cast<uint32_t*>(l.ob + LJ_T);
  l.list = reinterpret_cast<uint16_t*>(l.sm + lj_sync_sm_words(ns));
  l.misc = reinterpret_cast<uint32_t*>(l.list + LJ_T);
  return l;
}
// the byte after the layout (the stitch kernels' class tables)
__device__ __forceinline__ uint8_t* sync_lds_end(const Lds& l) {
  return reinterpret_cast<uint8_t*>(l.misc + 16);
}

// LDS addresses as integers (the dynamic LDS of these kernels starts at address 0)
typedef const __attribute__((address_space(3))) uint32_t* lds_u32p;
typedef const __attribute__((address_space(3))) uint16_t* lds_u16p;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return uint32_t(reinterpret_cast<uintptr_t>(
      (const __attribute__((address_space(3))) void*)(p)));
}
// Workgroup barrier that orders LDS accesses ONLY.  __syncthreads() is a workgroup-scope
// release + acquire over every address space, and on gfx9 the release waits for the
// acknowledgement of every global store (and the return of every load) the wavefront has in
// flight -- s_waitcnt vmcnt(0) in front of the s_barrier: a workgroup that has just written
// its slots out, or a guess, or a look-back record, stands still for a memory round trip at
// its next barrier although nobody in the workgroup will ever read those bytes.  Use where
// the lanes of a workgroup talk to one another through LDS alone.  (The compiler still waits
// for a loaded value in front of its first use: it counts the loads itself.)
#ifdef RSX_PLAIN_BARRIERS
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif
// the difference sums of slot j (N components)
template <int NS>
__device__ __forceinline__ uint2 sm_get(const Lds& L, int j) {
  if (NS == 0)
    return make_uint2(0u, 0u);
  if (NS <= 2)
    return make_uint2(L.sm[j], 0u);
  return make_uint2(L.sm[2 * j], L.sm[2 * j + 1]);
}
template <int NS>
__device__ __forceinline__ void sm_set(const Lds& L, int j, uint2 v) {
  if (NS == 0)
    return;
  if (NS <= 2) {
    L.sm[j] = v.x;
  } else {
    L.sm[2 * j] = v.x;
    L.sm[2 * j + 1] = v.y;
  }
}
static_assert(lj_lds_words(LJ_BW_SYNC) % 4 == 0 && lj_lds_words(LJ_BW_SYNC_PAIR) % 4 == 0 &&
                  lj_lds_words(LJ_BW) % 4 == 0,
              "the tables start on a 16-byte boundary");

__device__ __forceinline__ void lj_stage_tables(const Lds& L, const LjArgs& a,
                                                const LjStreamDev& S) {
  const uint4* src = reinterpret_cast<const uint4*>(a.tables + S.table_base);
  uint4* dst = reinterpret_cast<uint4*>(L.tabs);
  const int n16 = int(S.n_tables * sizeof(TabLds) / 16);
  for (int i = threadIdx.x; i < n16; i += int(blockDim.x))
    dst[i] = src[i];
}

// ... as TabLds10: every other entry of the 11-bit LUT (the two entries of a code of at
// most 10 bits are the same; an 11-bit code becomes "not in the LUT")
__device__ __forceinline__ void lj_stage_tables10(const Lds& L, const LjArgs& a,
                                                  const LjStreamDev& S) {
  static_assert(LUT_BITS == 11, "TabLds10 halves an 11-bit LUT");
  const TabLds* src = a.tables + S.table_base;
  TabLds10* dst = reinterpret_cast<TabLds10*>(L.tabs);
  for (uint32_t t = 0; t < S.n_tables; ++t) {
    for (int i = threadIdx.x; i < 1024; i += int(blockDim.x)) {
      const uint32_t e =
          reinterpret_cast<const uint16_t*>(src[t].lut)[2 * i * (sizeof(LutEntry) / 2)];
      dst[t].lut[i] = uint16_t((e & 31u) > 10u ? 0u : e);
    }
    // (max_code .. las: the same bytes in both structs)
    constexpr int tail = int(sizeof(TabLds10) - sizeof(dst[t].lut));
    static_assert(sizeof(TabLds) - sizeof(src[t].lut) == size_t(tail), "same tail");
    const uint32_t* ts = reinterpret_cast<const uint32_t*>(src[t].max_code);
    uint32_t* td = reinterpret_cast<uint32_t*>(dst[t].max_code);
    for (int i = threadIdx.x; i < tail / 4; i += int(blockDim.x))
      td[i] = ts[i];
  }
}

// End of the data the bit reader hands out before its zero padding.  An MSB32
// reader consumes whole little-endian words: the bytes of a partial last word are
// its LOW-order (= last) stream bits, so the data ends at the next word boundary.
__device__ __forceinline__ uint64_t lj_data_end(const LjStreamDev& S) {
  return S.pair ? (S.in_bytes + 3) & ~uint64_t(3) : S.in_bytes;
}

// Load the first BW dword rows of the workgroup's un-stuffed image (K0's output)
// into LDS, plus ob[].  REV: dword row k goes to LDS row BW-1-k -- the window reader
// of the synchronisation kernels then finds the two dwords of a window in the order
// of a 64-bit register pair (lj_window).  Ends with a workgroup barrier.
template <int BW = LJ_BW, bool REV = false>
__device__ __forceinline__ void lj_load_image(const Lds& L, const LjArgs& a, uint32_t b,
                                              int j) {
  const uint4* __restrict__ src = a.unstuffed + size_t(b) * LJ_IMG_U4;
  uint4* dst = reinterpret_cast<uint4*>(L.B);
  const uint32_t ob = reinterpret_cast<const uint32_t*>(src + (LJ_BW / 4) * LJ_T)[j];
  // the image is B as it lies in LDS ([dword][slot]); BW * LJ_T / 4 consecutive uint4
  // are wanted (a register array here ends up in scratch; copy in groups of three)
  constexpr int n4 = BW * LJ_T / 4;
  auto want = [&](int i) { return BW == LJ_BW || i < n4; };
  // uint4 i of the image = dwords 4 * (i % 64) .. of row i / 64
  auto at = [&](int i) { return REV ? (BW - 1 - (i >> 6)) * (LJ_T / 4) + (i & 63) : i; };
#pragma unroll
  for (int h = 0; h < LJ_BW / 4; h += 3) {
    const int i0 = h * LJ_T + j, i1 = i0 + LJ_T, i2 = i1 + LJ_T;
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0;
    if (want(i0))
      t0 = src[i0];
    if (h + 1 < LJ_BW / 4 && want(i1))
      t1 = src[i1];
    if (h + 2 < LJ_BW / 4 && want(i2))
      t2 = src[i2];
    if (want(i0))
      dst[at(i0)] = t0;
    if (h + 1 < LJ_BW / 4 && want(i1))
      dst[at(i1)] = t1;
    if (h + 2 < LJ_BW / 4 && want(i2))
      dst[at(i2)] = t2;
  }
  L.ob[j] = uint16_t(ob);
  __syncthreads();
}

// The 32 stream bits at bit position `pos` of slot `col`.  REVBW != 0: the image lies
// in LDS with its REVBW dword rows reversed (lj_load_image<REVBW, true>): dword wi + 1
// then sits one row BELOW dword wi, a ds_read2st64 delivers (d1, d0) as the low and
// high half of a register pair and the 64-bit shift needs no moves.
template <int REVBW = 0>
__device__ __forceinline__ uint32_t lj_window(const uint32_t* B, int col, uint32_t pos) {
  const uint32_t wi = pos >> 5;
  uint32_t d0, d1;
  if (REVBW) {
    const uint32_t r1 = uint32_t(REVBW - 2) - wi; // row of dword wi + 1
    d1 = B[r1 * LJ_T + col];
    d0 = B[(r1 + 1) * LJ_T + col];
  } else {
    d0 = B[wi * LJ_T + col];
    d1 = B[(wi + 1) * LJ_T + col];
  }
  return uint32_t((((uint64_t(d0) << 32) | d1) << (pos & 31u)) >> 32);
}

// Packed symbol entry (the LUT's format): bits 0..4 code length, 5..9 SSSS,
// 10..15 bits consumed.  0 = invalid code.
template <typename TB>
__device__ __noinline__ uint32_t lj_slow_entry(uint32_t w, const TB* tb) {
  // codes longer than the LUT: JPEG Annex F.2.2.3 search
  for (uint32_t l = TabBits<TB>::value + 1; l <= tb->max_len; ++l) {
    const uint32_t c = w >> (32 - l);
    const uint32_t mc = tb->max_code[l];
    if (mc != NO_CODE && c <= mc) {
      const uint32_t val = tb->values[(c - tb->val_offset[l]) & 0xFFFFu];
      const uint32_t ssss = (tb->las && val != 16u) ? (val & 15u) : val;
      const uint32_t extra = ssss == 16u ? (tb->fix16 ? 16u : 0u)
                                         : (tb->las ? ssss - (val >> 4) : ssss);
      return l | (ssss << 5) | ((l + extra) << 10);
    }
  }
  return 0u;
}

// Entry of the symbol whose first 32 bits are w.  `live` lanes matter; the
// out-of-line search only runs when some live lane missed the LUT (codes longer
// than LUT_BITS are rare, and never occur with the short tables real files use),
// so the common path has no divergent control flow at all.
// long_codes (wave-uniform): the stream's tables have codes longer than the LUT at all.
// The difference a symbol stands for (JPEG F.2.2.1 "EXTEND"; SSSS = 16 is -32768,
// AbstractPrefixCodeDecoder.h:55-76), as 16 bits: w = the symbol's window, e its entry.
__device__ __forceinline__ uint32_t lj_extend(uint32_t w, uint32_t e) {
  const uint32_t cl = e & 31u, ssss = (e >> 5) & 31u;
  const uint32_t x = w << cl;                    // the SSSS difference bits, top-aligned
  const uint32_t v = (x >> 1) >> (31u - ssss);   // (SSSS = 0: nothing)
  // a leading 0 bit means a negative difference: v - (2^SSSS - 1)
  const uint32_t all = (1u << ssss) - 1u;
  const uint32_t neg = ~uint32_t(int32_t(x) >> 31);
  uint32_t diff = v - (all & neg);
  diff = ssss == 16u ? 0x8000u : diff;
  return diff & 0xFFFFu;
}

// the 16-bit entry (code length | SSSS << 5 | bits consumed << 10) at LUT index i
__device__ __forceinline__ uint32_t lj_lut16(const TabLds& tb, uint32_t i) {
  return reinterpret_cast<const uint16_t*>(tb.lut)[i * (sizeof(LutEntry) / 2)];
}
__device__ __forceinline__ uint32_t lj_lut16(const TabLds10& tb, uint32_t i) {
  return tb.lut[i];
}

template <typename TB>
__device__ __forceinline__ uint32_t lj_entry(uint32_t w, const TB& tb, bool live,
                                             bool long_codes = true) {
  uint32_t e = lj_lut16(tb, w >> (32 - TabBits<TB>::value));
  if (long_codes && __builtin_expect(__any(live && (e & 31u) == 0u), 0)) {
    if (live && (e & 31u) == 0u)
      e = lj_slow_entry(w, &tb);
  }
  return e;
}

// Entry AND difference of the symbol whose first 32 bits are w (the loops that need
// both): *diff = the 16-bit difference.  With RSX_LUT_DIFF it comes out of the LUT's
// high half when the symbol lies inside the index bits; otherwise (wave-uniform branch:
// only when some live lane has a longer symbol) it is computed.
__device__ __forceinline__ uint32_t lj_entry_diff(uint32_t w, const TabLds10& tb, bool live,
                                                  bool long_codes, uint32_t* diff) {
  const uint32_t e = lj_entry(w, tb, live, long_codes);
  *diff = lj_extend(w, e);
  return e;
}
__device__ __forceinline__ uint32_t lj_entry_diff(uint32_t w, const TabLds& tb, bool live,
                                                  bool long_codes, uint32_t* diff) {
#if RSX_LUT_DIFF
  const uint32_t e32 = tb.lut[w >> (32 - LUT_BITS)];
  uint32_t e = e32 & 0xFFFFu;
  uint32_t d = e32 >> 16;
  const bool computed = live && ((e >> 10) > uint32_t(LUT_BITS) || e == 0u);
  if (__builtin_expect(__any(computed), 0)) {
    if (long_codes && live && (e & 31u) == 0u)
      e = lj_slow_entry(w, &tb);
    if (computed)
      d = lj_extend(w, e);
  }
  *diff = d;
  return e;
#else
  const uint32_t e = lj_entry(w, tb, live, long_codes);
  *diff = lj_extend(w, e);
  return e;
#endif
}

// the same for a single lane reading the table from global memory
__device__ __forceinline__ uint32_t lj_entry_global(uint32_t w, const TabLds* tb) {
  uint32_t e = lj_lut16(*tb, w >> (32 - LUT_BITS));
  if ((e & 31u) == 0u)
    e = lj_slow_entry(w, tb);
  return e;
}

// Per-stream decode parameters held in registers.
struct DecodeParams {
  uint32_t period;
  uint64_t tabmap; // byte p = table slot of phase p
  bool long_codes; // some table has codes longer than the LUT (default: assume so)
};

__device__ __forceinline__ DecodeParams lj_params(const LjStreamDev& S) {
  DecodeParams d;
  d.period = S.period;
  uint64_t m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    m |= uint64_t(S.tab_of_phase[i]) << (8 * i);
  d.tabmap = m;
  d.long_codes = true;
  return d;
}

// Register bit reader over column `col` of B.  The next 33..64 stream bits sit
// MSB-first in `buf`; the following dword is already prefetched in `nextw`, so the
// only LDS access on a symbol's critical path is the code-table lookup.
template <int BW = LJ_BW>
struct BitReader {
  uint64_t buf;
  uint32_t nb;    // valid bits in buf (33..64 between symbols)
  uint32_t wi;    // dword index of nextw
  uint32_t nextw;

  __device__ __forceinline__ void open(const uint32_t* B, int col, uint32_t pos) {
    const uint32_t i = pos >> 5, sh = pos & 31u;
    const uint32_t d0 = B[i * LJ_T + col], d1 = B[(i + 1) * LJ_T + col];
    buf = ((uint64_t(d0) << 32) | d1) << sh;
    nb = 64u - sh;
    wi = i + 2;
    nextw = B[wi * LJ_T + col];
  }
  // consume `len` bits (0 for a lane that must not advance) and top the buffer up
  __device__ __forceinline__ void advance(const uint32_t* B, int col, uint32_t len) {
    buf <<= len;
    nb -= len;
    const bool need = nb <= 32u;
    const uint64_t add = uint64_t(nextw) << ((32u - nb) & 31u);
    buf |= need ? add : 0ull;
    nb += need ? 32u : 0u;
    wi += need ? 1u : 0u;
    // past the slot's dwords there is nothing to read (a stopped lane may sit
    // there, and the bits a live symbol can still need end in dword LJ_PW);
    // clamp instead of branching
    const uint32_t w2 = wi < uint32_t(BW) ? wi : uint32_t(BW - 1);
    nextw = B[w2 * LJ_T + col];
  }
  __device__ __forceinline__ uint32_t head() const { return uint32_t(buf >> 32); }
};

// whether any of the stream's n tables (staged in LDS) has codes longer than the LUT
// (wave-uniform)
template <typename TB = TabLds>
__device__ __forceinline__ bool lj_long_codes(const Lds& L, uint32_t n_tables) {
  uint32_t m = 0;
  for (uint32_t t = 0; t < n_tables; ++t)
    m = max(m, uint32_t(reinterpret_cast<const TB*>(L.tabs)[t].max_len));
  return __builtin_amdgcn_readfirstlane(int(m)) > TabBits<TB>::value;
}

template <bool MULTI, typename TB = TabLds>
__device__ __forceinline__ const TB& lj_table(const Lds& L, const DecodeParams& dp,
                                              uint32_t phase) {
  return reinterpret_cast<const TB*>(L.tabs)[MULTI ? uint32_t(dp.tabmap >> (8 * phase)) & 0xFFu
                                                   : 0u];
}

// ---- packed 16-bit arithmetic (predictors wrap mod 2^16) ---------------------
typedef unsigned short rsx_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t x, uint32_t y) {
  const rsx_u16x2 r = __builtin_bit_cast(rsx_u16x2, x) + __builtin_bit_cast(rsx_u16x2, y);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t x, uint32_t y) {
  const rsx_u16x2 r = __builtin_bit_cast(rsx_u16x2, x) - __builtin_bit_cast(rsx_u16x2, y);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint2 pk_add2(uint2 x, uint2 y) {
  return make_uint2(pk_add(x.x, y.x), pk_add(x.y, y.y));
}
__device__ __forceinline__ uint2 pk_sub2(uint2 x, uint2 y) {
  return make_uint2(pk_sub(x.x, y.x), pk_sub(x.y, y.y));
}

// Four 16-bit fields q = 0..3 (x: q0 | q1 << 16, y: q2 | q3 << 16), of which the
// first N are in use.  Rotation: out[c] = in[(c - f) mod N], i.e. field q moves to
// q + f.  Sums kept by RELATIVE phase (k mod N for the k-th symbol of a run whose
// first symbol has absolute index `first`) become sums by absolute phase with
// f = first mod N; f = N - (first mod N) undoes it.
template <int N>
__device__ __forceinline__ uint2 lj_rot_fields(uint2 v, uint32_t f) {
  if (N == 1)
    return v;
  if (N == 3) {
    // three fields of a 48-bit word (field 3 stays 0): f = 1: (f2, f0, f1), f = 2: (f1, f2, f0)
    const uint64_t w = (uint64_t(v.x) | (uint64_t(v.y) << 32)) & 0xFFFFFFFFFFFFull;
    const uint64_t r = f == 0u ? w
                               : (f == 1u ? ((w << 16) | (w >> 32)) : ((w << 32) | (w >> 16))) &
                                     0xFFFFFFFFFFFFull;
    return make_uint2(uint32_t(r), uint32_t(r >> 32));
  }
  if (N == 2) {
    const uint32_t x = (f & 1u) ? __builtin_amdgcn_alignbit(v.x, v.x, 16) : v.x;
    return make_uint2(x, 0u);
  }
  const uint64_t w = uint64_t(v.x) | (uint64_t(v.y) << 32);
  const uint32_t sh = 16u * (f & 3u);
  const uint64_t r = sh ? ((w << sh) | (w >> (64u - sh))) : w;
  return make_uint2(uint32_t(r), uint32_t(r >> 32));
}

// Accumulator of differences by relative phase (N = 1, 2 or 4 components).
template <int N>
struct PhaseSums {
  uint32_t a0 = 0, a1 = 0;
  uint32_t ph = 0; // 16 * (symbols so far mod N)
  __device__ __forceinline__ void add(uint32_t d16, bool live) {
    const uint32_t d = live ? d16 : 0u;
    if (N == 1) {
      a0 += d;
    } else if (N == 2) {
      a0 = pk_add(a0, d << ph);
      ph = live ? ph ^ 16u : ph;
    } else {
      const uint32_t t = d << (ph & 16u);
      a0 = pk_add(a0, (ph & 32u) ? 0u : t);
      a1 = pk_add(a1, (ph & 32u) ? t : 0u);
      ph = live ? (ph + 16u) & 63u : ph;
    }
  }
  __device__ __forceinline__ uint2 get() const {
    return make_uint2(N == 1 ? (a0 & 0xFFFFu) : a0, N == 4 ? a1 : 0u);
  }
};

} // namespace rsx
