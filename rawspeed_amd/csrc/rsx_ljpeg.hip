// rsx_ljpeg.hip -- lossless-JPEG family decode pipeline for gfx950: un-stuffing,
// synchronisation, scans and the host-side plan.  (The fused decode +
// reconstruction lives in rsx_ljpeg_direct.hip, the legacy reconstruction from
// differences and the Nikon / Pentax / Sony kernels in rsx_ljpeg_recon.hip;
// shared structures in rsx_ljpeg_dev.h, device building blocks in rsx_ljpeg_bits.h.)
//
// Replaces the serial loops of
//   LJpegDecompressor::decodeN / decodeRowN  (decompressors/LJpegDecompressor.cpp:184-339)
//   Cr2Decompressor::decompressN_X_Y         (decompressors/Cr2DecompressorImpl.h:396-468)
//   NikonDecompressor::decompress            (decompressors/NikonDecompressor.cpp:515-560)
//   PentaxDecompressor::decompress           (decompressors/PentaxDecompressor.cpp:152-176)
// which walk ONE bit streamer over the whole scan (no per-row index).
//
// Pipeline (one "stream" = one scan or one restart interval; many streams per
// launch: DNG tiles, batched frames).  The physical byte stream is cut into
// 64-byte subsequences ("slots"), one lane each, 256 per workgroup; slot 0 of a
// workgroup is a copy of the previous workgroup's last slot.
//
//  K0 lj_unstuff   every lane loads its slot (+16 bytes of lookahead) into
//                  registers and parks it big-endian in its LDS column.  Slots that
//                  hold an FF are collected in a dense list and un-stuffed by the
//                  first lanes (FF00 -> FF, FFxx / end of buffer end the data).
//                  The LDS image is written to global memory once.
//  K1 lj_sync      self-synchronising speculative Huffman decode: lane j decodes
//                  the tail of slot j-1 from an arbitrary bit to find where its own
//                  slot most likely starts, decodes the slot -- recording exit
//                  state, symbol count and, for the fused path, the sums of its
//                  differences by relative component phase -- and the workgroup
//                  iterates "re-decode from the predecessor's exit state" on a
//                  dense list of the few slots that guessed wrong.
//  K2 lj_sync<STITCH>  cross-workgroup fix-up: a workgroup whose assumed start
//                  differs from its predecessor's recorded exit re-converges.
//                  (Jacobi iteration: a fixed point is the serial decode.)
//  K3 lj_scan      per stream: verify the chain, exclusive scans of the symbol
//                  counts and of the difference sums (-> running sums P of every
//                  component before each workgroup).
//  --  lj_transfer / lj_chain   fallback for streams that do not self-synchronise
//                  (periodic data): per-workgroup transfer functions over all
//                  entry states, chained by one lane per stream.
//  then either (rsx_ljpeg_direct.hip: LJPEG / CR2 streams with 1, 2 or 4
//  interleaved components)
//  K5a lj_rowedge  P at the first MCU of every stream row
//  K5b lj_rowoff   row offsets O(r, c): X(i) = P(i) + O(row(i), comp(i))
//  K4d lj_decode_direct  final decode from validated start states straight into
//                  the image: no difference scratch, no second pass
//  or (every other stream kind, and damaged streams)
//  K4 lj_decode    final decode into a stream-ordered int16 scratch
//  K4b lj_tail     exact end-of-stream semantics for damaged streams
//  K5 / K6         seeds + row scans (rsx_ljpeg_recon.hip)
//  and
//  K7 lj_consumed  decode()'s return value (SURVEY.md A.6 closed form).
//
// Symbol semantics: codes/AbstractPrefixCodeDecoder.h:43-76; end-of-stream:
// bitstreams/BitStreamerJPEG.h:106-183.  No MFMA (no contraction anywhere).
#include "rsx_ljpeg.h"
#include "rsx_ljpeg_bits.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <type_traits>

namespace rsx {

namespace {

// ---------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t has_ff(uint32_t d) {
  return ((~d) - 0x01010101u) & d & 0x80808080u;
}

// 16 bytes at stream offset `off`, zero outside [0, in_bytes)
__device__ __forceinline__ uint4 lj_load_chunk(const uint8_t* __restrict__ base,
                                               int64_t off, int64_t in_bytes,
                                               bool aligned16) {
  if (off >= 0 && off + 16 <= in_bytes && aligned16)
    return *reinterpret_cast<const uint4*>(base + off);
  uint32_t w[4] = {0, 0, 0, 0};
  if (off < in_bytes && off + 16 > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t o = off + i;
      const uint32_t b = (o >= 0 && o < in_bytes) ? base[o] : 0u;
      w[i >> 2] |= b << (8 * (i & 3));
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- un-stuffing of one slot, branch-free -----------------------------------
// exact per-byte "== 0" flags of a dword (0x80 in every zero byte)
__device__ __forceinline__ uint32_t lj_zero_flags(uint32_t d) {
  return ~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d | 0x7F7F7F7Fu);
}
// flags at bits 31/23/15/7 (stream bytes 0..3 of a big-endian dword) -> bit j = byte j
__device__ __forceinline__ uint32_t lj_flag_nibble(uint32_t m) {
  return ((m >> 31) | (m >> 22) | (m >> 13) | (m >> 4)) & 0xFu;
}
// v_perm selector that moves the bytes kept by `nib` (bit j = stream byte j of a
// big-endian dword) to the top of the result and zero-fills the rest
__device__ __forceinline__ uint32_t lj_compact_selector(uint32_t nib) {
  uint32_t sel = 0, o = 0;
#pragma unroll
  for (uint32_t j = 0; j < 4; ++j)
    if (nib & (1u << j)) {
      sel |= (4u + 3u - j) << (8u * (3u - o)); // stream byte j = register byte 3 - j
      ++o;
    }
  for (; o < 4; ++o)
    sel |= 0x0Cu << (8u * (3u - o)); // constant 0x00
  return sel;
}

// Un-stuff one slot held in registers (20 big-endian dwords = its 64 bytes + 16
// bytes of lookahead) into column `col` of B.  FF00 -> FF; FFxx (xx != 0) or the
// end of the buffer end the data, everything after reads as zero
// (BitStreamerJPEG.h:106-183).  `valid` = bytes of the slot that lie inside the
// buffer, `prev` = the byte before the slot, `sel` = 16 compaction selectors.
//
// Formulated on 80-bit byte masks (bit i = stream byte i of the slot) so that
// the wavefront runs ONE instruction stream whatever the lanes hold -- the
// byte-walking version this replaces took ~7000 instructions per wavefront as
// soon as the lanes' FF bytes sat in different dwords:
//   FF, Z   bytes equal to FF / 00          V      bytes inside the buffer
//   M  = FF & V & ((V & ~Z) >> 1)           FF followed by a non-zero byte: a marker
//   E  = first bit of M, else `valid`       the data of this slot ends here
//   D  = Z & (FF << 1 | prev == FF) & [0,E) stuffing bytes: dropped
//   K  = [0,E) & ~D                          kept; compacted dword by dword (v_perm)
__device__ __forceinline__ void lj_fix_regs(const uint32_t (&in)[LJ_BW + 1], uint32_t* B,
                                            int col, uint32_t prev, int valid,
                                            const uint32_t* sel, uint32_t& own_bits,
                                            int& marker_off, uint32_t& own_drops) {
  static_assert(LJ_BW == 20 && LJ_PW == 16, "80-bit masks: 64 own + 16 lookahead bytes");
  uint64_t ffl = 0, zl = 0; // bytes 0..63
  uint32_t ffh = 0, zh = 0; // bytes 64..79
#pragma unroll
  for (int k = 0; k < LJ_BW; ++k) {
    const uint32_t nz = lj_flag_nibble(lj_zero_flags(in[k]));
    const uint32_t nf = lj_flag_nibble(lj_zero_flags(~in[k]));
    if (k < LJ_PW) {
      zl |= uint64_t(nz) << (4 * k);
      ffl |= uint64_t(nf) << (4 * k);
    } else {
      zh |= nz << (4 * (k - LJ_PW));
      ffh |= nf << (4 * (k - LJ_PW));
    }
  }
  const uint32_t nv = uint32_t(valid); // 0..80
  const uint64_t vl = nv >= 64u ? ~0ull : ((1ull << nv) - 1ull);
  const uint32_t vh = nv <= 64u ? 0u : ((1u << (nv - 64u)) - 1u);
  // byte i + 1 exists and is not zero
  const uint64_t nzl = ((vl & ~zl) >> 1) | (uint64_t(vh & ~zh & 1u) << 63);
  const uint32_t nzh = (vh & ~zh) >> 1;
  const uint64_t ml = ffl & vl & nzl;
  const uint32_t mh = ffh & vh & nzh;
  const bool marker = (ml != 0ull) || (mh != 0u);
  const uint32_t E = ml ? uint32_t(__builtin_ctzll(ml))
                        : (mh ? 64u + uint32_t(__builtin_ctz(mh)) : nv);
  const uint64_t bl = E >= 64u ? ~0ull : ((1ull << E) - 1ull);
  const uint32_t bh = E <= 64u ? 0u : ((1u << (E - 64u)) - 1u);
  const uint64_t pfl = (ffl << 1) | (prev == 0xFFu ? 1ull : 0ull);
  const uint32_t pfh = (ffh << 1) | uint32_t(ffl >> 63);
  const uint64_t dl = zl & pfl & bl;
  const uint32_t dh = zh & pfh & bh;
  const uint64_t kl = bl & ~dl;
  const uint32_t kh = bh & ~dh;
  own_bits = 8u * uint32_t(__builtin_popcountll(kl));
  own_drops = uint32_t(__builtin_popcountll(dl));
  marker_off = (marker && E < 64u) ? int(E) : -1;

  uint64_t acc = 0;  // kept bytes, top-aligned; the high `nacc` bits are valid
  uint32_t nacc = 0; // 0, 8, 16 or 24 between dwords
  uint32_t ko = 0;   // output dwords written
#pragma unroll
  for (int k = 0; k < LJ_BW; ++k) {
    const uint32_t nib = k < LJ_PW ? uint32_t(kl >> (4 * k)) & 0xFu
                                   : (kh >> (4 * (k - LJ_PW))) & 0xFu;
    const uint32_t x = __builtin_amdgcn_perm(in[k], 0u, sel[nib]);
    acc |= (uint64_t(x) << 32) >> nacc;
    nacc += 8u * uint32_t(__builtin_popcount(nib));
    if (nacc >= 32u) {
      B[ko * LJ_T + col] = uint32_t(acc >> 32);
      ++ko;
      acc <<= 32;
      nacc -= 32u;
    }
  }
  if (ko < uint32_t(LJ_BW)) {
    B[ko * LJ_T + col] = uint32_t(acc >> 32);
    ++ko;
  }
#pragma unroll 1
  for (; ko < uint32_t(LJ_BW); ++ko)
    B[ko * LJ_T + col] = 0u;
}

// phase time stamps of K0 in experiment builds (second half of LjArgs::dbg; printed next to
// the single-pass kernel's by ljpeg_plan_results under RSX_DEBUG)
#ifdef RSX_EXPERIMENT
#define K0_STAMP(k)                                                                   \
  do {                                                                                \
    if (a.dbg && threadIdx.x == 0)                                                    \
      a.dbg[(size_t(a.n_blocks_plan) + (S.first_block + lb)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define K0_STAMP(k) \
  do {              \
  } while (0)
#endif

__device__ __forceinline__ int lj_valid_bytes(const LjStreamDev& S, uint32_t lb, int j) {
  const int64_t vb =
      int64_t(lj_data_end(S)) - (int64_t(lb) * LJ_R + int64_t(j - 1) * LJ_P);
  return vb < 0 ? 0 : (vb > 4 * LJ_BW ? 4 * LJ_BW : int(vb));
}

// Stage the workgroup's 256 slots: every lane loads its slot (64 bytes + 16
// bytes of lookahead) straight from global memory and parks it big-endian in
// column j of B.  Slots that hold an FF anywhere (~27 % of them) need
// un-stuffing; they are collected in a dense list and fixed by the first lanes
// of the workgroup, so the rare byte-level work runs at full lane utilisation
// instead of dragging every wavefront through it.  Slot j of workgroup lb holds
// stream bytes [lb*LJ_R + (j-1)*LJ_P, +LJ_P).
// Outputs (LDS): B, ob[] (own data bits per slot), misc[9] (stuffing bytes
// dropped in owned slots); the end marker goes to results[s].marker_pos.
// Ends with a workgroup barrier.
__device__ __forceinline__ void lj_stage_slots(const Lds& L, const LjArgs& a,
                                               const LjStreamDev& S, uint32_t s,
                                               uint32_t lb, int j, bool report_marker) {
  const uint8_t* __restrict__ in = a.in_base + S.in_offset;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
  const int64_t in_bytes = int64_t(S.in_bytes);
  const int64_t start = int64_t(lb) * LJ_R + int64_t(j - 1) * LJ_P;
  uint4 v[LJ_BW / 4];
  // A stream that does not start on the buffer's 16-byte grid -- every restart interval
  // (it starts two bytes behind a marker), a DNG tile at an odd offset: all its slots are
  // off the grid by the SAME delta, so a lane takes the six grid chunks its 80 bytes lie in
  // and funnels them down by delta bytes (a workgroup-uniform amount).  Until round 5 such
  // streams were assembled byte by byte, eighty loads a lane: K0 took 0.18 ms instead of
  // 0.07 on the four tiles of a cfg-4 frame with restart intervals.
  const uint32_t delta = uint32_t(reinterpret_cast<uintptr_t>(in) & 15u);
#ifdef RSX_NO_FUNNEL // (A/B builds)
  const bool funnel = false;
#else
  const bool funnel = !aligned16 && S.in_offset >= 16u && start >= 0 &&
                      start - int64_t(delta) + 96 <= in_bytes;
#endif
  if (__any(funnel)) {
    uint32_t w[25];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const uint4 c = funnel ? *reinterpret_cast<const uint4*>(in + (start - int64_t(delta)) + 16 * k)
                             : make_uint4(0, 0, 0, 0);
      w[4 * k] = c.x, w[4 * k + 1] = c.y, w[4 * k + 2] = c.z, w[4 * k + 3] = c.w;
    }
    w[24] = 0;
    const uint32_t q = uint32_t(__builtin_amdgcn_readfirstlane(int(delta >> 2)));
    const uint32_t r = uint32_t(__builtin_amdgcn_readfirstlane(int(delta & 3u)));
    uint32_t dd[LJ_BW];
#pragma unroll
    for (int i = 0; i < LJ_BW; ++i) {
      // (q is workgroup-uniform: four static register choices under uniform branches)
      const uint32_t lo = q == 0 ? w[i] : (q == 1 ? w[i + 1] : (q == 2 ? w[i + 2] : w[i + 3]));
      const uint32_t hi = q == 0 ? w[i + 1] : (q == 1 ? w[i + 2] : (q == 2 ? w[i + 3] : w[i + 4]));
      dd[i] = __builtin_amdgcn_alignbyte(hi, lo, r);
    }
#pragma unroll
    for (int m = 0; m < LJ_BW / 4; ++m)
      v[m] = funnel ? make_uint4(dd[4 * m], dd[4 * m + 1], dd[4 * m + 2], dd[4 * m + 3])
                    : lj_load_chunk(in, start + 16 * m, in_bytes, aligned16);
  } else {
#pragma unroll
    for (int m = 0; m < LJ_BW / 4; ++m)
      v[m] = lj_load_chunk(in, start + 16 * m, in_bytes, aligned16);
  }
  const uint32_t prev =
      (start >= 1 && start - 1 < in_bytes) ? uint32_t(in[start - 1]) : 0u;
  if (j == 0) {
    L.misc[9] = 0;  // dropped stuffing bytes
    L.misc[10] = 0; // fix-list length
  }
  if (j < 16) // sm[] is free during staging: the 16 byte-compaction selectors
    L.sm[j] = lj_compact_selector(uint32_t(j));
  uint32_t any = 0;
#pragma unroll
  for (int m = 0; m < LJ_BW / 4; ++m) {
    // MSB32 (Hasselblad): a little-endian word already is the next 32 stream bits
    const bool le = S.pair != 0;
    const uint32_t d[4] = {le ? v[m].x : __builtin_bswap32(v[m].x),
                           le ? v[m].y : __builtin_bswap32(v[m].y),
                           le ? v[m].z : __builtin_bswap32(v[m].z),
                           le ? v[m].w : __builtin_bswap32(v[m].w)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      any |= has_ff(d[q]);
      L.B[(4 * m + q) * LJ_T + j] = d[q];
    }
  }
  const int valid = lj_valid_bytes(S, lb, j);
  L.ob[j] = 8u * uint32_t(valid > LJ_P ? LJ_P : valid);
  L.su[j] = prev; // su[] doubles as the "byte before the slot" array during staging
  __syncthreads();
  if (report_marker)
    K0_STAMP(1);
  if ((any != 0u || prev == 0xFFu) && !S.raw)
    L.list[atomicAdd(&L.misc[10], 1u)] = uint16_t(j);
  __syncthreads();
  if (report_marker)
    K0_STAMP(2);
  const uint32_t n = L.misc[10];
  if (uint32_t(j & ~63) < n) { // wave-uniform
    const bool mine = uint32_t(j) < n;
    const int idx = mine ? int(L.list[j]) : 0;
    uint32_t r[LJ_BW + 1];
#pragma unroll
    for (int k = 0; k < LJ_BW; ++k)
      r[k] = L.B[k * LJ_T + idx];
    r[LJ_BW] = 0;
    if (mine) {
      uint32_t own_bits, drops;
      int marker_off;
      lj_fix_regs(r, L.B, idx, L.su[idx], lj_valid_bytes(S, lb, idx), L.sm, own_bits,
                  marker_off, drops);
      L.ob[idx] = own_bits;
      if (idx >= 1) {
        if (drops)
          atomicAdd(&L.misc[9], drops);
        if (report_marker && marker_off >= 0) {
          const int64_t p = int64_t(lb) * LJ_R + int64_t(idx - 1) * LJ_P + marker_off;
          if (p >= 0)
            atomicMin(&a.results[s].marker_pos, uint32_t(p));
        }
      }
    }
  }
  __syncthreads();
}


// Entry of the symbol that starts at `pos` (0 = invalid code); *w_out = its window.
// REVBW: layout of the image (lj_window).
template <bool MULTI, bool PAIR, int REVBW, typename TB = TabLds>
__device__ __forceinline__ uint32_t lj_step(const Lds& L, const DecodeParams& dp, int col,
                                            uint32_t pos, uint32_t phase, bool live,
                                            uint32_t* w_out = nullptr,
                                            uint32_t* d_out = nullptr) {
  const uint32_t w = lj_window<REVBW>(L.B, col, pos);
  if (w_out)
    *w_out = w;
  const TB& tb = lj_table<MULTI, TB>(L, dp, phase);
  uint32_t e;
  if (d_out) // (the loops that accumulate differences)
    e = lj_entry_diff(w, tb, live, dp.long_codes, d_out);
  else
    e = lj_entry(w, tb, live, dp.long_codes);
  if (PAIR) {
    // HasselbladDecompressor.cpp:87-92: two length codes, then the two bit fields.
    // The step covers the whole pair: advance = both codes + both fields (<= 64).
    const uint32_t e2 = lj_entry(lj_window<REVBW>(L.B, col, pos + (e & 31u)), tb,
                                 live && e != 0u, dp.long_codes);
    return (e != 0u && e2 != 0u) ? ((((e >> 10) + (e2 >> 10)) << 10) | 1u) : 0u;
  }
  return e;
}

// Decode the symbols that START inside slot `col` (bit positions [.., end_bits)),
// beginning at state `start`.  NS != 0: *sums receives the sums of the differences
// by relative phase (k mod NS for the k-th symbol).  All lanes of the wave run the
// same loop; a lane that is done simply stops advancing (no divergent branches in
// the hot loop).  With one shared table the component phase does not influence the
// parse, so it is left out of the state (it would never self-synchronise); with
// several tables it is part of what has to match.
// The same loop for the common case -- one table, plain symbols, the reversed image --
// with the address arithmetic spelled out (this loop is what the synchronisation kernel
// IS: 100 % of the VALU issue slots are busy while it runs, so every instruction per
// symbol is 1.4 % of the kernel).  LDS addresses are integers: the window's row comes out
// of one multiply-add, the LUT entry's address out of an OR (the table sits at LDS
// address 0), the symbol count is an add-with-carry on the "good" mask, and the phase
// of a difference is its symbol's number.
template <int NS, int REVBW, typename TB>
__device__ __forceinline__ void lj_decode_span_single(const Lds& L, const DecodeParams& dp,
                                                      int col, uint32_t start,
                                                      uint32_t end_bits, uint32_t& exit,
                                                      uint32_t& count, uint2* sums,
                                                      bool enabled, uint32_t pos_override) {
  static_assert(REVBW != 0, "the reversed image");
  constexpr int LB = TabBits<TB>::value;
  uint32_t pos = pos_override != 0xFFFFFFFFu ? pos_override : (start & ST_OFF_MASK);
  bool ok = !(start & ST_ERR);
  if (!ok || !enabled)
    end_bits = 0; // lane takes no steps
  const TB& tb = *reinterpret_cast<const TB*>(L.tabs);
  const uint32_t vrow = lds_addr(&L.B[(REVBW - 2) * LJ_T + col]); // dword 1 of the slot
  const uint32_t lut = lds_addr(tb.lut);
  uint32_t n = 0, a0 = 0, a1 = 0;
  bool live = pos < end_bits;
  if (__any(live)) {
    do {
      // dword wi + 1 lies wi rows below `vrow`, dword wi one row further down
      // (one multiply-add: left to itself the compiler makes a shift, an AND and a subtract of it)
      uint32_t ad;
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(ad) : "v"(pos >> 5), "s"(-4 * LJ_T), "v"(vrow));
      const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
      const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << (pos & 31u)) >> 32);
      uint32_t e = *(lds_u16p)(lut | ((w >> (31 - LB)) & ((2u << LB) - 2u)));
      if (dp.long_codes && __builtin_expect(__any(live && (e & 31u) == 0u), 0)) {
        if (live && (e & 31u) == 0u)
          e = lj_slow_entry(w, &tb);
      }
      const bool good = live && e != 0u;
      if (NS) {
        const uint32_t dx = lj_extend(w, e);
        const uint32_t d = good ? dx : 0u;
        const uint32_t t = d << ((n << 4) & 31u); // (phase = symbol number: 16 * (n & 1))
        if (NS <= 2) {
          a0 = NS == 1 ? a0 + d : pk_add(a0, t);
        } else {
          a0 = pk_add(a0, (n & 2u) ? 0u : t);
          a1 = pk_add(a1, (n & 2u) ? t : 0u);
        }
      }
      pos += good ? (e >> 10) : 0u;
      n += good ? 1u : 0u;
      if (live && !good) {
        ok = false;
        end_bits = 0;
      }
      live = pos < end_bits;
    } while (__any(live));
  }
  if (!enabled)
    return;
  exit = ok ? (pos - end_bits) : ST_ERR;
  count = n;
  if (NS)
    *sums = make_uint2(NS == 1 ? (a0 & 0xFFFFu) : a0, NS == 4 ? a1 : 0u);
}

#ifndef RSX_SPAN_GENERIC
#define RSX_SPAN_GENERIC 0 // (experiments: 1 = the generic loop everywhere)
#endif
template <bool MULTI, int NS, bool PAIR, int REVBW, typename TB = TabLds>
__device__ __forceinline__ void lj_decode_span(const Lds& L, const DecodeParams& dp,
                                               int col, uint32_t start,
                                               uint32_t end_bits, uint32_t& exit,
                                               uint32_t& count, uint2* sums,
                                               bool enabled = true,
                                               uint32_t pos_override = 0xFFFFFFFFu) {
  if (!MULTI && !PAIR && REVBW != 0 && !RSX_SPAN_GENERIC) {
    lj_decode_span_single<NS, REVBW ? REVBW : 1, TB>(L, dp, col, start, end_bits, exit, count,
                                                     sums, enabled, pos_override);
    return;
  }
  // pos_override: start at an arbitrary bit position of the slot (warm-up)
  uint32_t pos = pos_override != 0xFFFFFFFFu ? pos_override : (start & ST_OFF_MASK);
  uint32_t phase = (start >> ST_PHASE_SHIFT) & 7u;
  uint32_t n = 0;
  bool ok = !(start & ST_ERR);
  if (!ok || !enabled)
    end_bits = 0; // lane takes no steps
  PhaseSums<NS ? NS : 1> acc;
  while (__any(pos < end_bits)) {
    const bool live = pos < end_bits;
    uint32_t w, d = 0;
    const uint32_t e =
        lj_step<MULTI, PAIR, REVBW, TB>(L, dp, col, pos, phase, live, &w, NS ? &d : nullptr);
    const bool bad = live && e == 0u;
    const bool good = live && !bad;
    if (NS)
      acc.add(d, good);
    pos += good ? (e >> 10) : 0u;
    n += good ? 1u : 0u;
    if (MULTI)
      phase = good ? ((phase + 1 == dp.period) ? 0u : phase + 1) : phase;
    if (bad) {
      ok = false;
      end_bits = 0;
    }
  }
  if (!enabled)
    return;
  exit = ok ? ((pos - end_bits) | (MULTI ? (phase << ST_PHASE_SHIFT) : 0u)) : ST_ERR;
  count = n;
  if (NS)
    *sums = acc.get();
}

// Start-state guess for slot j: decode the last LJ_WARM bits of slot j-1 from an
// arbitrary bit position; Huffman streams self-synchronise within a few
// symbols, so the position at which this runs into slot j is almost always the
// true one.  (Checked against the predecessor's real exit afterwards.)
template <bool MULTI, bool PAIR, int REVBW, typename TB = TabLds>
__device__ __forceinline__ uint32_t lj_warmup(const Lds& L, const DecodeParams& dp,
                                              int j) {
  // j == 0 has no predecessor slot in LDS: it takes no steps (`enabled` false)
  const bool enabled = j >= 1;
  const uint32_t prev_bits = enabled ? L.ob[j - 1] : 0u;
  const uint32_t from = prev_bits > LJ_WARM ? prev_bits - LJ_WARM : 0u;
  uint32_t e = 0, c = 0;
  lj_decode_span<MULTI, 0, PAIR, REVBW, TB>(L, dp, enabled ? j - 1 : 0, 0u, prev_bits, e, c,
                                            nullptr, enabled && prev_bits != 0, from);
  return (e & ST_ERR) ? 0u : e;
}

// ---------------------------------------------------------------------------
// K0: un-stuffing.  Every workgroup stages its 256 slots (lj_stage_slots) and
// writes the un-stuffed LDS image -- B plus the per-slot data-bit counts -- to
// global memory once; the synchronisation, stitch and decode kernels start from
// that image with plain 16-byte coalesced loads.  Keeping the byte-level work in
// its own light kernel (25 KB LDS) hides its latency.
// ---------------------------------------------------------------------------
// Start guesses for the single-pass kernel (rsx_ljpeg_fast.hip): where the parse of a
// slot from bit `from` ends, as an offset into the next slot.  lut8 = total bits of the
// symbol a 10-bit window starts with.
// (LDS addresses by hand: left to itself the compiler keeps two base pointers for the two
// dwords of the window and spends four instructions on their addresses; this way the row
// offset is an AND and a shift-add and the pair is one ds_read2st64_b32)
typedef const __attribute__((address_space(3))) uint8_t* lds_u8p;
template <bool COUNT>
__device__ __forceinline__ uint32_t lj_guess_parse(uint32_t bcol, uint32_t lut, uint32_t end_bits,
                                                   uint32_t from, uint32_t* count = nullptr) {
  static_assert(LJ_T == 256, "row stride 1 KB = 32 bits << 5");
  uint32_t pos = from, n = 0, spec = 0;
  while (pos < end_bits) {
    const uint32_t ad = bcol + ((pos & ~31u) << 5);
    const uint32_t d0 = *(lds_u32p)(ad), d1 = *(lds_u32p)(ad + 4u * LJ_T);
    const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << (pos & 31u)) >> 32);
    const uint32_t len = *(lds_u8p)(lut + (w >> 22));
    spec |= len;
    pos += len & 0x7Fu;
    if (COUNT)
      ++n;
  }
  if (COUNT)
    *count = n | ((spec & 0x80u) << 24);
  return pos - end_bits;
}
// The same loop instruction by instruction, for the layout the kernel really has (the
// image at LDS address 0, the length table behind the layout): the position is kept as
// 32 * pos, so the window's row is one v_and_or, its two dwords come with one
// ds_read2st64_b32 in the order v_lshlrev_b64 wants them, the advance is one v_lshl_add --
// 6 vector instructions and 2 LDS reads per symbol where the compiler's version has 11 and 2
// (K0 parses three slots per slot: 0.20 of its 0.33 ms on cfg 3 are this loop).
constexpr uint32_t LJ_GUESS_LUT_OFF = uint32_t(LJ_PW + 1) * LJ_T * 4; // dword row 17 of the image
constexpr uint32_t LJ_K0_OFF_OB = uint32_t(LJ_BW) * LJ_T * 4;
constexpr uint32_t LJ_K0_LDS = LJ_K0_OFF_OB + 3 * LJ_T * 2 + 16 * 4 + 16 * 4 + 16;
template <bool COUNT>
__device__ __forceinline__ uint32_t lj_guess_parse_asm(uint32_t col4, uint32_t end_bits,
                                                       uint32_t from, uint32_t* count = nullptr) {
  uint32_t q = from << 5, n = 0, spec = 0;
  const uint32_t qend = end_bits << 5;
  while (q < qend) {
    uint32_t ad, len;
    uint64_t pr;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    // (two reads: 56 cycles where ds_read2st64_b32 takes 73, scripts/ubench/valu_rates.hip)
    uint32_t d0, d1;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(ad));
    pr = (uint64_t(d0) << 32) | d1;
    const uint32_t w = uint32_t((pr << ((q >> 5) & 31u)) >> 32);
    asm volatile("ds_read_u8 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(len)
                 : "v"(w >> 22), "n"(LJ_GUESS_LUT_OFF));
    spec |= len;
    q += (len & 0x7Fu) << 5;
    if (COUNT)
      ++n;
  }
  if (COUNT)
    *count = n | ((spec & 0x80u) << 24);
  return (q - qend) >> 5;
}

// Total bits of the symbol a window starts with, the general way (any code length, SSSS = 16;
// 0: no such code): what the guess parses fall back to where the 10-bit length table only
// says "special".  Rare -- the codes of 11 bits and more are the rarest categories -- but a
// count the table merely approximates makes the whole workgroup "uncertain", and every
// workgroup behind it in flight then waits for its decode (two tables with long codes:
// 0.2 % of the workgroups, 8 us of the AVERAGE workgroup's 40).  In the two-table parse only:
// the test per symbol cost the one-table loops 14 % of K0 (cfg 3: 0.255 -> 0.29 ms).
__device__ __forceinline__ uint32_t lj_exact_symbol_bits(uint32_t w, const TabLds* tb) {
  uint32_t r = lj_lut16(*tb, w >> (32 - LUT_BITS));
  if ((r & 31u) == 0u) {
    r = 0;
    for (uint32_t l = LUT_BITS + 1; l <= tb->max_len && r == 0u; ++l) {
      const uint32_t c = w >> (32 - l);
      const uint32_t mc = tb->max_code[l];
      if (mc != NO_CODE && c <= mc) {
        const uint32_t ssss = tb->values[(c - tb->val_offset[l]) & 0xFFFFu];
        const uint32_t extra = ssss == 16u ? (tb->fix16 ? 16u : 0u) : ssss;
        r = (l + extra) << 10;
      }
    }
  }
  return r >> 10;
}

// Two symbols per window read (round 4).  The loop above spends, per symbol, two LDS reads
// for the window, one random byte read in a 1 KB table (four dwords per bank: conflicts)
// and ~7 vector instructions.  Here a second, 256-byte table -- 64 dwords, one per LDS bank:
// conflict-free -- holds the symbol's total length for the codes of up to 8 bits (0x80:
// longer code, special entry), and ONE 32-bit window serves two look-ups: the second
// symbol's code starts at bit l1 of it, and l1 <= 24 for every entry of the small table (a
// code of at most 8 bits + at most 16 difference bits), so the 8 index bits are the stream's.
// Pairs are taken while the first symbol cannot reach the end of the slot (q + 26 symbols'
// bits < end), so the second one always starts inside it; the last two or three symbols go
// through the one-symbol loop.  A miss in either look-up (0x80 + anything >= 128) takes ONE
// symbol by the 10-bit table, as the loop above would.
// 10 vector instructions and 4 LDS reads per PAIR where the loop above takes 14-16 and 6.
#ifndef RSX_K0_FINAL_POLLS
#define RSX_K0_FINAL_POLLS 64 // polls for the predecessor's FINAL hand-over word (two-table plans)
#endif
constexpr uint32_t LJ_GUESS_ROUNDS = 6; // rounds of the chain's fixed-point iteration, at most
// dword rows 17..19 of the image (3 KB, free once the image is written out): the 10-bit
// length table | the 8-bit one, or the SECOND 10-bit table of a two-table stream | the chain's
// two state arrays.  The rounds' slot list and their words live in the staging's list and
// selector arrays, which are free by then: the second table costs no LDS, and K0 keeps its
// seven workgroups a CU.
constexpr uint32_t LJ_GUESS_LUT8_OFF = LJ_GUESS_LUT_OFF + 1024u;
constexpr uint32_t LJ_GUESS_EA_OFF = LJ_GUESS_LUT_OFF + 2048u;
static_assert((4 + 2 * LJ_GUESS_ROUNDS) * 4 <= 64, "the rounds' words fit the selector array");
static_assert(LJ_GUESS_EA_OFF + 1024u <= uint32_t(LJ_BW) * LJ_T * 4, "inside dword rows 17..19");
template <bool COUNT>
__device__ __forceinline__ uint32_t lj_guess_parse_pairs(uint32_t col4, uint32_t end_bits,
                                                         uint32_t from, uint32_t* count = nullptr) {
  uint32_t q = from << 5, n = 0, spec = 0;
  const uint32_t qend = end_bits << 5;
  // (unsigned: a slot shorter than 27 bits has no pair region at all)
  const uint32_t qpair = qend > (26u << 5) ? qend - (26u << 5) : 0u;
  while (q < qpair) {
    uint32_t ad;
    uint64_t pr;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    // (dword row + 1 into the LOW register, dword row into the high one: the pair the
    // 64-bit shift wants, without moves)
    asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:0\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(pr)
                 : "v"(ad));
    const uint32_t w = uint32_t((pr << ((q >> 5) & 31u)) >> 32);
    const uint32_t l1 = *(lds_u8p)(LJ_GUESS_LUT8_OFF + (w >> 24));
    // (a hit is a code of at most 8 bits + at most 16 difference bits: l1 <= 24, so the
    // second symbol's 8 index bits are the stream's; a miss shifts by 0x80 & 31 = 0 and
    // the sum says so)
    const uint32_t w2 = w << (l1 & 31u);
    const uint32_t l2 = *(lds_u8p)(LJ_GUESS_LUT8_OFF + (w2 >> 24));
    const uint32_t sum = l1 + l2;
    // (the pair's advance first, overridden on a miss: ONE skipped block per iteration -- as
    // an if / else the compiler gave each arm a block of its own and the loop three taken
    // branches an iteration: K0 0.245 -> 0.275 ms)
    uint32_t qadd = sum << 5, nadd = 2u;
    if (__builtin_expect(sum >= 128u, 0)) {
      const uint32_t len = *(lds_u8p)(LJ_GUESS_LUT_OFF + (w >> 22));
      spec |= len;
      qadd = (len & 0x7Fu) << 5;
      nadd = 1u;
    }
    q += qadd;
    if (COUNT)
      n += nadd;
  }
  // the rest, symbol by symbol
  while (q < qend) {
    uint32_t ad;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    const uint32_t d0 = *(lds_u32p)(ad), d1 = *(lds_u32p)(ad + 4u * LJ_T);
    const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << ((q >> 5) & 31u)) >> 32);
    const uint32_t len = *(lds_u8p)(LJ_GUESS_LUT_OFF + (w >> 22));
    spec |= len;
    q += (len & 0x7Fu) << 5;
    if (COUNT)
      ++n;
  }
  // (bit 31 of the count: the parse met an entry the 10-bit table only approximates -- a
  // code of more than 10 bits, SSSS = 16, a hole in the code space: bit 7 of its length)
  if (COUNT)
    *count = n | ((spec & 0x80u) << 24);
  return (q - qend) >> 5;
}

// Two tables that alternate symbol by symbol (round 4; LjStreamDev::fast == 2): the table of
// the next symbol is part of the parse state -- offset | (symbol index & 1) << 6.  The second
// table's lengths take the 8-bit table's place; the two 8-bit tables of the pair loop lie
// behind the kernel's usual layout, in what the allocation granule leaves unused anyway.
constexpr uint32_t LJ_K0_LUTB_OFF = LJ_GUESS_LUT8_OFF;
constexpr uint32_t LJ_K0_LUT8A_OFF = (LJ_K0_LDS + 15u) & ~15u, LJ_K0_LUT8B_OFF = LJ_K0_LUT8A_OFF + 256u;
constexpr uint32_t LJ_K0_LDS_MT = LJ_K0_LUT8B_OFF + 256u;
static_assert(LJ_K0_LDS_MT <= 18u * 1280u, "seven workgroups a CU: 18 granules of LDS each");
constexpr uint32_t ST_MT_MASK = 0x7Fu; // offset | table bit
// (EXACT: with the general look-up behind every "special" length -- for the parses that start
// from a predecessor's exit.  The parse from bit 0 runs through garbage until it falls into
// step, meets such an entry in one slot of sixteen there, and a wavefront waits for the
// slowest of its lanes: looked up there too, every wavefront paid a round trip to the
// table in global memory for lengths nobody needs.)
template <bool COUNT, bool EXACT>
__device__ __forceinline__ uint32_t lj_guess_parse_mt(uint32_t col4, uint32_t end_bits,
                                                      uint32_t from, const TabLds* tb_even,
                                                      const TabLds* tb_odd,
                                                      uint32_t* count = nullptr) {
  uint32_t q = (from & ST_OFF_MASK) << 5, n = 0, spec = 0;
  // the tables of the NEXT symbol (10-bit, 8-bit): a single symbol swaps them with the other
  // pair (XOR with the bases' difference), a pair of symbols leaves them as they are
  uint32_t lut = (from & 64u) ? LJ_K0_LUTB_OFF : LJ_GUESS_LUT_OFF;
  uint32_t l8 = (from & 64u) ? LJ_K0_LUT8B_OFF : LJ_K0_LUT8A_OFF;
  const uint32_t qend = end_bits << 5;
  const uint32_t qpair = qend > (26u << 5) ? qend - (26u << 5) : 0u;
  while (q < qpair) { // (lj_guess_parse_pairs' loop, the second look-up in the other table)
    uint32_t ad;
    uint64_t pr;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:0\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(pr)
                 : "v"(ad));
    const uint32_t w = uint32_t((pr << ((q >> 5) & 31u)) >> 32);
    const uint32_t l1 = *(lds_u8p)(l8 + (w >> 24));
    const uint32_t w2 = w << (l1 & 31u);
    const uint32_t l2 = *(lds_u8p)((l8 ^ (LJ_K0_LUT8A_OFF ^ LJ_K0_LUT8B_OFF)) + (w2 >> 24));
    const uint32_t sum = l1 + l2;
    uint32_t qadd = sum << 5, nadd = 2u;
    if (__builtin_expect(sum >= 128u, 0)) {
      uint32_t len = *(lds_u8p)(lut + (w >> 22));
      if (EXACT && __builtin_expect(len & 0x80u, 0)) {
        const uint32_t exact = lj_exact_symbol_bits(w, lut == LJ_K0_LUTB_OFF ? tb_odd : tb_even);
        if (exact)
          len = exact;
      }
      spec |= len;
      qadd = (len & 0x7Fu) << 5;
      nadd = 1u;
      lut ^= LJ_K0_LUTB_OFF ^ LJ_GUESS_LUT_OFF;
      l8 ^= LJ_K0_LUT8A_OFF ^ LJ_K0_LUT8B_OFF;
    }
    q += qadd;
    n += nadd;
  }
  while (q < qend) {
    uint32_t ad;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    const uint32_t d0 = *(lds_u32p)(ad), d1 = *(lds_u32p)(ad + 4u * LJ_T);
    const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << ((q >> 5) & 31u)) >> 32);
    uint32_t len = *(lds_u8p)(lut + (w >> 22));
    if (EXACT && __builtin_expect(len & 0x80u, 0)) {
      const uint32_t exact = lj_exact_symbol_bits(w, lut == LJ_K0_LUTB_OFF ? tb_odd : tb_even);
      if (exact)
        len = exact;
    }
    spec |= len;
    q += (len & 0x7Fu) << 5;
    lut ^= LJ_K0_LUTB_OFF ^ LJ_GUESS_LUT_OFF;
    ++n;
  }
  if (COUNT)
    *count = n | ((spec & 0x80u) << 24);
  return ((q - qend) >> 5) | (lut == LJ_K0_LUTB_OFF ? 64u : 0u);
}

// A table per PHASE (round 6; LjStreamDev::fast == 3): the table of a symbol is the one of its
// index mod N (N = 2, 3, 4 components, up to four tables in any assignment) -- the parse state is
// offset | phase << 6.  The N 10-bit length tables and the N 8-bit ones of the pair loop lie behind
// the kernel's usual layout (5 KB: five workgroups a CU for these plans instead of seven).
// (NP = the most phases a stream of the plan has, LjArgs::pt_np: three phases leave a CU six
// workgroups, four leave it five)
constexpr uint32_t LJ_K0_PT10_OFF = (LJ_K0_LDS + 15u) & ~15u;      // NP x 1024 B
__host__ __device__ constexpr uint32_t lj_k0_pt8_off(uint32_t np) { return LJ_K0_PT10_OFF + np * 1024u; } // NP x 256 B
__host__ __device__ constexpr uint32_t lj_k0_ptx_off(uint32_t np) { return lj_k0_pt8_off(np) + np * 256u; } // 8 words: the phase pass
// ... and the words of its rounds: phases fall into step more slowly than offsets (a parse in the
// wrong phase meets the true one at the same bit AND in the same phase once in N times), so a
// table-per-phase stream's chain gets twelve rounds where the others get six -- with twelve every
// workgroup of tests/test_per_phase_model.py's streams settles, with six half of those with four tables
constexpr uint32_t LJ_GUESS_ROUNDS_PT = 12;
__host__ __device__ constexpr uint32_t lj_k0_ptr_off(uint32_t np) { return lj_k0_ptx_off(np) + 32u; } // 2 words a round
__host__ __device__ constexpr uint32_t lj_k0_lds_pt(uint32_t np) { return lj_k0_ptr_off(np) + 8u * LJ_GUESS_ROUNDS_PT; }
// K0's look-back over symbol counts mod N (LjArgs::k0p): a workgroup's word is its own total
// (AGG) as soon as its chain has settled, then the phase it ENDS in (INC) once it knows where it starts
constexpr uint32_t K0P_AGG = 1u << 30, K0P_INC = 1u << 31;
constexpr uint32_t ST_PT_MASK = 0xFFu; // offset | phase
template <bool COUNT, bool EXACT>
__device__ __forceinline__ uint32_t lj_guess_parse_pt(uint32_t col4, uint32_t end_bits, uint32_t from,
                                                      uint32_t np, const TabLds* tabs0,
                                                      uint32_t tabsel, uint32_t pt8,
                                                      uint32_t* count = nullptr) {
  uint32_t q = (from & ST_OFF_MASK) << 5, n = 0, spec = 0;
  uint32_t ph = (from >> ST_PHASE_SHIFT) & 3u; // phase of the NEXT symbol
  const uint32_t qend = end_bits << 5;
  const uint32_t qpair = qend > (26u << 5) ? qend - (26u << 5) : 0u;
  auto next = [np](uint32_t p) -> uint32_t { return p + 1u == np ? 0u : p + 1u; };
  while (q < qpair) { // (lj_guess_parse_pairs' loop, the second look-up in the next phase's table)
    uint32_t ad;
    uint64_t pr;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:0\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(pr)
                 : "v"(ad));
    const uint32_t w = uint32_t((pr << ((q >> 5) & 31u)) >> 32);
    const uint32_t ph1 = next(ph);
    const uint32_t l1 = *(lds_u8p)(pt8 + (ph << 8) + (w >> 24));
    const uint32_t w2 = w << (l1 & 31u);
    const uint32_t l2 = *(lds_u8p)(pt8 + (ph1 << 8) + (w2 >> 24));
    const uint32_t sum = l1 + l2;
    uint32_t qadd = sum << 5, nadd = 2u, phn = next(ph1);
    if (__builtin_expect(sum >= 128u, 0)) {
      uint32_t len = *(lds_u8p)(LJ_K0_PT10_OFF + (ph << 10) + (w >> 22));
      if (EXACT && __builtin_expect(len & 0x80u, 0)) {
        const uint32_t exact = lj_exact_symbol_bits(w, tabs0 + ((tabsel >> (4u * ph)) & 15u));
        if (exact)
          len = exact;
      }
      spec |= len;
      qadd = (len & 0x7Fu) << 5;
      nadd = 1u;
      phn = ph1;
    }
    q += qadd;
    n += nadd;
    ph = phn;
  }
  while (q < qend) {
    uint32_t ad;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ad) : "v"(q), "s"(0xFFFFFC00u), "v"(col4));
    const uint32_t d0 = *(lds_u32p)(ad), d1 = *(lds_u32p)(ad + 4u * LJ_T);
    const uint32_t w = uint32_t((((uint64_t(d0) << 32) | d1) << ((q >> 5) & 31u)) >> 32);
    uint32_t len = *(lds_u8p)(LJ_K0_PT10_OFF + (ph << 10) + (w >> 22));
    if (EXACT && __builtin_expect(len & 0x80u, 0)) {
      const uint32_t exact = lj_exact_symbol_bits(w, tabs0 + ((tabsel >> (4u * ph)) & 15u));
      if (exact)
        len = exact;
    }
    spec |= len;
    q += (len & 0x7Fu) << 5;
    ph = next(ph);
    ++n;
  }
  if (COUNT)
    *count = n | ((spec & 0x80u) << 24);
  return ((q - qend) >> 5) | (ph << ST_PHASE_SHIFT);
}

// A slot inside a constant region of the image is the code of the zero difference over
// and over.  No parse from an arbitrary bit finds its way into such a stretch reliably
// (with Nikon's 14-bit table, 111110 repeated also reads as a chain of 12-bit symbols),
// but the BITS say where its symbols start: if the slot has period `zl` and the code `zc`
// at exactly one phase p, the first symbol boundary behind the slot follows.  A guess like
// any other: the single-pass kernel checks it against the predecessor's exit.
// Two alternating tables (zlb != 0): the slot is the two zero codes in turn; zl, zc are the
// pair's (first table's code, then the second's, zlb bits), the states carry the table bit.
__device__ __forceinline__ bool lj_guess_constant(const uint32_t* B, int col, uint32_t zl,
                                                  uint32_t zc, bool candidate, uint32_t* guess,
                                                  uint32_t* count, uint32_t* entry,
                                                  uint32_t zlb, uint32_t bits) {
  // (the slot's own bits -- fewer than 512 when it held stuffing bytes: two zero codes in
  // turn easily make a run of eight ones -- and up to 31 + zl of the bits behind them, which
  // the staging put there: at least 88 of the next slot's)
  bool per = candidate && bits >= 64u;
  for (int wi = 0; wi < LJ_PW && __any(per); ++wi) {
    const uint32_t d0 = B[wi * LJ_T + col], d1 = B[(wi + 1) * LJ_T + col];
    per = per && (uint32_t(32 * wi) >= bits ||
                  d0 == uint32_t((((uint64_t(d0) << 32) | d1) << zl) >> 32));
  }
  if (!per)
    return false;
  const uint64_t w64 = (uint64_t(B[col]) << 32) | B[LJ_T + col];
  uint32_t hits = 0, p0 = 0;
  for (uint32_t p = 0; p < zl; ++p)
    if (uint32_t((w64 << p) >> (64u - zl)) == zc) {
      ++hits;
      p0 = p;
    }
  if (hits != 1)
    return false;
  if (zlb != 0u) {
    // symbols of the first table start at p0 + k zl, of the second at sb + k zl
    const uint32_t zla = zl - zlb;
    const bool b_first = p0 >= zlb;
    const uint32_t sb = b_first ? p0 - zlb : p0 + zla;
    const uint32_t na = (bits - p0 + zl - 1u) / zl, nb = (bits - sb + zl - 1u) / zl;
    const uint32_t end_a = p0 + na * zl, end_b = sb + nb * zl; // the first starts behind the slot
    *guess = end_a < end_b ? end_a - bits : ((end_b - bits) | 64u);
    *count = na + nb;
    *entry = b_first ? (sb | 64u) : p0;
    return true;
  }
  const uint32_t r = (bits - p0) % zl;
  *guess = r ? zl - r : 0u;
  *count = (bits - p0 + zl - 1u) / zl;
  *entry = p0; // the symbol grid of the slot: its first symbol starts at bit p0
  return true;
}
// ... and for a table per phase: the slot is the N zero codes in turn, period zl = the sum of their
// lengths, zc = the codes one behind the other (phase 0's first), cum[k] = where phase k's code starts
// inside the pattern.  The pattern sits at exactly one bit p0 of the period; symbols of phase k then
// start at every bit = p0 + cum[k] (mod zl).  Entry = the first of them in the slot, exit = the
// first behind it, count = all in between.
__device__ __forceinline__ bool lj_guess_constant_pt(const uint32_t* B, int col, uint32_t zl, uint32_t zc,
                                                     const uint32_t (&cum)[4], uint32_t np,
                                                     bool candidate, uint32_t* guess, uint32_t* count,
                                                     uint32_t* entry, uint32_t bits) {
  bool per = candidate && bits >= 64u;
  for (int wi = 0; wi < LJ_PW && __any(per); ++wi) {
    const uint32_t d0 = B[wi * LJ_T + col], d1 = B[(wi + 1) * LJ_T + col];
    per = per && (uint32_t(32 * wi) >= bits ||
                  d0 == uint32_t((((uint64_t(d0) << 32) | d1) << zl) >> 32));
  }
  if (!per)
    return false;
  const uint64_t w64 = (uint64_t(B[col]) << 32) | B[LJ_T + col];
  uint32_t hits = 0, p0 = 0;
  for (uint32_t p = 0; p < zl; ++p)
    if (uint32_t((w64 << p) >> (64u - zl)) == zc) {
      ++hits;
      p0 = p;
    }
  if (hits != 1)
    return false;
  uint32_t best_in = 0xFFFFFFFFu, best_out = 0xFFFFFFFFu, n = 0;
  for (uint32_t k = 0; k < np; ++k) {
    uint32_t f = p0 + cum[k];
    f = f >= zl ? f - zl : f; // first start of a phase-k symbol in the slot (p0, cum[k] < zl)
    if (f >= bits)
      continue; // (cannot happen: bits >= 64 > zl)
    const uint32_t m = (bits - f + zl - 1u) / zl; // phase-k symbols that start inside the slot
    n += m;
    const uint32_t out = f + m * zl - bits; // its first start behind the slot
    if ((f << 2 | k) < best_in)
      best_in = f << 2 | k;
    if ((out << 2 | k) < best_out)
      best_out = out << 2 | k;
  }
  *entry = (best_in >> 2) | ((best_in & 3u) << ST_PHASE_SHIFT);
  *guess = (best_out >> 2) | ((best_out & 3u) << ST_PHASE_SHIFT);
  *count = n;
  return true;
}
constexpr int LJ_GUESS_SLOTS = 3; // slots parsed for a guess, at most (LjArgs::guess_slots)

// MTPLAN: the plan has streams with two alternating tables.  Two instantiations because the
// mere presence of their code -- the two-table parse, the hand-over of entry states between
// workgroups -- made the kernel 4-10 % slower for plans that never run it (register
// allocation and layout of the rounds: cfg 3 0.258 -> 0.265-0.29 ms).
// KM: 0 plans of one-table streams, 1 plans with two-alternating-table streams, 2 plans with
// table-per-phase streams (whose two-table streams are table-per-phase streams as well)
#define K0_CHAIN (MTPLAN && a.k0_chain != 0u)
template <int KM, bool INV>
__global__ __launch_bounds__(LJ_T) void lj_unstuff_kernel(LjArgs a) {
  constexpr bool MTPLAN = KM != 0;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // The LAST workgroups first: the kernels behind this one read the un-stuffed image from
  // its first workgroup on, and what went through the 256 MB memory-side cache last is
  // what they find there (cfg 3: the image alone is 343 MB; in block order the cache holds
  // its END when the readers start at its beginning -- measured: K0 -3 %, the single-pass
  // kernel -1 %, a single cfg-4 frame -2.8 %).  K0's workgroups are independent of one another.
  // (K0_CHAIN, plans with two-table streams: workgroups in block order, because each one asks
  // its predecessor for its true entry state, see "hand-over" below)
#ifdef RSX_K0_FORWARD
  uint32_t b = blockIdx.x;
#else
  uint32_t b = a.blk0 + (a.blk_n - 1u - blockIdx.x);
#endif
  // (no tickets: the dispatcher starts a 1-D grid's workgroups in order, and if it ever did
  // not, the bounded wait below gives up and the workgroup keeps its own estimate -- a ticket
  // and its barrier in front of the first load cost every workgroup 1.5 us of its 25)
  if (K0_CHAIN)
    b = a.blk0 + blockIdx.x;
  lj_fresh_scalars<INV>();
  const uint32_t s = a.block_stream[b];
  // The NEXT run's results (marker_pos = 0xFFFFFFFF, an atomicMin target; everything else 0),
  // a dword a lane of the first workgroups: the plan keeps two sets and takes turns, so a
  // run finds its set clean without a kernel in front of it (lj_init_results_kernel: 6 us in
  // front of every run, 3 % of a single cfg-4 frame; still launched for a plan's first run).
  {
    constexpr uint32_t DW = uint32_t(sizeof(LjResult) / 4);
    const uint32_t i = blockIdx.x * uint32_t(LJ_T) + threadIdx.x;
    if (a.results_next && i < a.n_streams * DW)
      reinterpret_cast<uint32_t*>(a.results_next)[i] =
          (i % DW == uint32_t(offsetof(LjResult, marker_pos) / 4)) ? 0xFFFFFFFFu : 0u;
  }
  if (s == 0xFFFFFFFFu)
    return; // (a block no stream owns: plans laid out on the device, lj_dri_layout_kernel)
  const LjStreamDev& S = a.streams[s];
  // This kernel's own layout: the arrays of the general one that only the synchronisation
  // kernels use are left out, and the length table of the start guesses goes where dword
  // rows 17..19 of the image were once the image is written out -- 18 LDS granules, a
  // SEVENTH workgroup on a CU (the guesses are bound by LDS latency: 0.315 -> 0.29 ms).
  Lds L{};
  L.B = reinterpret_cast<uint32_t*>(smem);
  L.ob = reinterpret_cast<uint16_t*>(smem + LJ_K0_OFF_OB);
  L.su = L.ob + LJ_T;
  L.list = L.su + LJ_T;
  L.sm = reinterpret_cast<uint32_t*>(L.list + LJ_T);
  L.misc = L.sm + 16;
  uint32_t* est = L.misc + 16; // symbols of the workgroup (estimate)
  const uint32_t lb = b - S.first_block;
  const int j = threadIdx.x;
  uint8_t* lut8 = smem + LJ_GUESS_LUT_OFF;
  uint32_t lut_pk = 0, lut8b = 0x80u, lut_pk_b = 0, lut8b_b = 0x80u;
  const bool mt = KM == 1 && S.fast == 2;
  const bool pt = KM == 2 && S.fast == 3;
  const uint32_t np = pt ? S.tab_period : 1u; // (phases of a table-per-phase stream)
  const uint32_t tabsel = pt ? (uint32_t(S.tab_of_phase[0] & 15u) | (uint32_t(S.tab_of_phase[1] & 15u) << 4) |
                                (uint32_t(S.tab_of_phase[2] & 15u) << 8) |
                                (uint32_t(S.tab_of_phase[3] & 15u) << 12))
                             : 0u;
  uint32_t pt_pk[4] = {0, 0, 0, 0}, pt_8[4] = {0x80u, 0x80u, 0x80u, 0x80u};
  if (KM == 2 && pt && a.fast_tabs) {
    // (the symbol lengths of every phase's 10-bit LUT: entries 4j .. 4j + 3, and the 8-bit
    // table's entry j, as below)
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
      if (k >= np)
        continue;
      const uint2* ft = a.fast_tabs + size_t(S.table_base + ((tabsel >> (4u * k)) & 15u)) * 1024 + 4 * j;
      uint2 e0 = make_uint2(0u, 0u);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 e = ft[q];
        if (q == 0)
          e0 = e;
        pt_pk[k] |= (((e.x >> 5) & 63u) | ((e.x >> 24) & 0x80u)) << (8 * q);
      }
      const uint32_t total = (e0.x >> 5) & 63u;
      const uint32_t code = total - uint32_t(__builtin_popcount(e0.y));
      if (!(e0.x & 0x80000000u) && code <= 8u && total >= 1u)
        pt_8[k] = total;
    }
  } else if (S.fast && a.fast_tabs) {
    // (the symbol lengths of the stream's 10-bit LUT: asked for now, parked later)
    const uint2* ft =
        a.fast_tabs + size_t(S.table_base + (mt ? S.tab_of_phase[0] : 0u)) * 1024 + 4 * j;
    uint2 e0 = make_uint2(0u, 0u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint2 e = ft[k];
      if (k == 0)
        e0 = e;
      lut_pk |= (((e.x >> 5) & 63u) | ((e.x >> 24) & 0x80u)) << (8 * k); // (bit 7: special)
    }
    if (mt) {
      const uint2* fb = a.fast_tabs + size_t(S.table_base + S.tab_of_phase[1]) * 1024 + 4 * j;
      uint2 b0 = make_uint2(0u, 0u);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint2 e = fb[k];
        if (k == 0)
          b0 = e;
        lut_pk_b |= (((e.x >> 5) & 63u) | ((e.x >> 24) & 0x80u)) << (8 * k);
      }
      const uint32_t total = (b0.x >> 5) & 63u;
      const uint32_t code = total - uint32_t(__builtin_popcount(b0.y));
      if (!(b0.x & 0x80000000u) && code <= 8u && total >= 1u)
        lut8b_b = total;
    }
    // the 8-bit table's entry j: the four 10-bit entries 4j.. agree iff the code has at most
    // 8 bits (code length = total - SSSS, SSSS = popcount of the entry's 2^SSSS - 1)
    const uint32_t total = (e0.x >> 5) & 63u;
    const uint32_t code = total - uint32_t(__builtin_popcount(e0.y));
    if (!(e0.x & 0x80000000u) && code <= 8u && total >= 1u)
      lut8b = total;
  }
  K0_STAMP(0);
  lj_stage_slots(L, a, S, s, lb, j, true); // ends with a barrier
  K0_STAMP(3);
  uint4* __restrict__ dst = a.unstuffed + size_t(b) * LJ_IMG_U4;
  const uint4* src = reinterpret_cast<const uint4*>(L.B);
#pragma unroll
  for (int m = 0; m < LJ_BW / 4; ++m)
    dst[m * LJ_T + j] = src[m * LJ_T + j];
  reinterpret_cast<uint32_t*>(dst + (LJ_BW / 4) * LJ_T)[j] = L.ob[j];
  if (j == 0)
    a.block_drops[b] = L.misc[9];
  // state of the single-pass kernel that follows (rsx_ljpeg_fast.hip): its look-back
  // granules and workgroup tickets start from zero in every run
  if (a.lb && j < LF_LB_WORDS)
    a.lb[size_t(b) * LF_LB_WORDS + j] = 0ull;
  if (a.tickets && b == 0 && j < 25) {
    if (j < 24)
      a.tickets[j] = 0u;
    else
      a.fast_level[a.run_parity ^ 1u] = a.fast_level[2u + (a.run_parity ^ 1u)] = 0u; // (the NEXT run's)
  }
  // Start guesses of the single-pass kernel: where does the parse that starts at bit 0 three
  // slots earlier run into slot t?  (Huffman streams self-synchronise: one slot leaves 1.7 %
  // of the guesses wrong, two 0.03 %, three next to none.)  Here and not in that kernel: a
  // wrong guess costs it a re-decode round, and every workgroup behind the re-decoding one
  // waits for its symbol count (measured with two slots parsed there: 9 % of the workgroups
  // re-decode, and the average workgroup waits 26 us for its predecessors).
  // Lane c parses slot c, each time from where the same chain left slot c - 1:
  //   A[c] = parse(c, 0),  B[c] = parse(c, A[c-1]),  then rounds  X[c] = parse(c, X'[c-1])
  // for the slots whose last parse did not start where their predecessor's last parse ended
  // (2 % after B; they go through a dense list on the first wavefront) until nothing changes:
  // a fixed point of the workgroup's chain -- every slot parsed from exactly the exit of the
  // slot before it, given slot 0's.  2.25 parses a slot, and since round 4 the slots'
  // SYMBOL COUNTS under those very entries, which are the entries the single-pass kernel
  // decodes from: the kernel takes its first symbol's index from the counts of the workgroups
  // before it (a sum it reads when it starts) instead of waiting for their decodes in a
  // look-back (3.6 us of a workgroup's 30).  What K0 cannot know -- slot 0's true exit is the
  // predecessor workgroup's business, codes the 10-bit table does not cover, data that does
  // not synchronise -- is marked, and the kernel asks those workgroups (and only those) for
  // the difference (rsx_ljpeg_fast.hip, "symbol base").
  if (S.fast && a.fast_tabs) {
    uint16_t* EA = reinterpret_cast<uint16_t*>(smem + LJ_GUESS_EA_OFF); // (dword row 19)
    uint16_t* EB = EA + LJ_T;
    uint16_t* glist = L.list; // (the staging's list of slots with stuffing bytes: done with)
    // [2] uncertain; [4 + 2 r] list length of round r, [5 + 2 r] "an exit moved in round r"
    uint32_t* nlist = L.sm; // (the staging's 16 compaction selectors: done with)
    uint16_t* EU = EA;      // (after the B parses) the entry a slot's last parse started from
    uint16_t* ECNT = L.su;  // symbols of a slot's last parse (su[] is the staging's)
    lds_barrier(); // every lane has written its part of the image out
    K0_STAMP(4);
    reinterpret_cast<uint32_t*>(lut8)[j] = lut_pk;
    if (KM == 2 && pt) {
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        if (k >= np)
          continue;
        reinterpret_cast<uint32_t*>(smem + LJ_K0_PT10_OFF + (k << 10))[j] = pt_pk[k];
        smem[lj_k0_pt8_off(a.pt_np) + (k << 8) + uint32_t(j)] = uint8_t(pt_8[k]);
      }
    } else if (mt) {
      reinterpret_cast<uint32_t*>(smem + LJ_K0_LUTB_OFF)[j] = lut_pk_b;
      smem[LJ_K0_LUT8A_OFF + uint32_t(j)] = uint8_t(lut8b);
      smem[LJ_K0_LUT8B_OFF + uint32_t(j)] = uint8_t(lut8b_b);
    } else {
      smem[LJ_GUESS_LUT8_OFF + uint32_t(j)] = uint8_t(lut8b);
    }
    if (j < 4 + 2 * int(LJ_GUESS_ROUNDS))
      nlist[j] = 0;
    // (the rounds' words: [2 r] list length of round r, [2 r + 1] "an exit moved in round r")
    uint32_t* const rw = KM == 2 ? reinterpret_cast<uint32_t*>(smem + lj_k0_ptr_off(a.pt_np)) : nlist + 4;
    const uint32_t max_rounds = (KM == 2 && pt) ? LJ_GUESS_ROUNDS_PT : LJ_GUESS_ROUNDS;
    if (KM == 2 && j < 2 * int(LJ_GUESS_ROUNDS_PT))
      rw[j] = 0;
    if (j == 0)
      *est = 0;
    lds_barrier();
    // (two tables: such a slot is the two zero codes in turn, a "code" of both lengths)
    uint32_t zi = uint32_t(__builtin_amdgcn_readfirstlane(
        int(a.fast_z[S.table_base + (mt ? S.tab_of_phase[0] : 0u)])));
    uint32_t zl = zi & 31u, zc = zi >> 8, zlb = 0;
    uint32_t zcum[4] = {0, 0, 0, 0};
    if (pt) {
      // (the N zero codes in turn: period = the sum of their lengths, the codes one behind the other)
      zl = 0u;
      zc = 0u;
      bool all = true;
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        if (k >= np)
          continue;
        const uint32_t zk = uint32_t(__builtin_amdgcn_readfirstlane(
            int(a.fast_z[S.table_base + ((tabsel >> (4u * k)) & 15u)])));
        const uint32_t lk = zk & 31u;
        all = all && lk != 0u;
        zcum[k] = zl;
        zl += lk;
        zc = (zc << lk) | (zk >> 8);
      }
      // (at most 128 symbols a slot: four bits a symbol on average; the pattern inside 31 bits)
      if (!all || zl > 31u || zl < 4u * np)
        zl = 0u;
    }
    if (mt) {
      const uint32_t zb = uint32_t(
          __builtin_amdgcn_readfirstlane(int(a.fast_z[S.table_base + S.tab_of_phase[1]])));
      zlb = zb & 31u;
      zc = (zc << zlb) | (zb >> 8);
      zl = (zl != 0u && zlb != 0u && zl + zlb <= 31u) ? zl + zlb : 0u;
    }
    const uint32_t smask = pt ? ST_PT_MASK : (mt ? ST_MT_MASK : ST_OFF_MASK);
    const uint32_t gs = a.guess_slots & 0xFFu; // (3; experiments: fewer)
    // (the hand-written loop assumes the layout it was written for)
    bool hand = lds_addr(L.B) == 0u && lds_addr(lut8) == LJ_GUESS_LUT_OFF;
#ifdef RSX_EXPERIMENT
    if (a.guess_slots & 0x100u) // (experiments: the compiler's loop)
      hand = false;
#endif
    // (the stream's tables in global memory: for the symbols the length tables only flag)
    const TabLds* tb_even = a.tables + S.table_base + (mt ? S.tab_of_phase[0] : 0u);
    const TabLds* tb_odd = a.tables + S.table_base + (mt ? S.tab_of_phase[1] : 0u);
    // (two tables, the parse from bit 0: special lengths as the table approximates them)
    const TabLds* tabs0 = a.tables + S.table_base;
    const uint32_t pt8 = lj_k0_pt8_off(a.pt_np); // (wave-uniform: a kernel argument)
    auto parse_from_0 = [&](int col, uint32_t bits, uint32_t* count) -> uint32_t {
      if (KM == 2)
        return lj_guess_parse_pt<true, false>(uint32_t(col) * 4u, bits, 0u, np, tabs0, tabsel, pt8, count);
      return lj_guess_parse_mt<true, false>(uint32_t(col) * 4u, bits, 0u, tb_even, tb_odd, count);
    };
    auto parse = [&](int col, uint32_t bits, uint32_t from, uint32_t* count) -> uint32_t {
      if (KM == 2 && pt)
        return count ? lj_guess_parse_pt<true, true>(uint32_t(col) * 4u, bits, from, np, tabs0, tabsel, pt8, count)
                     : lj_guess_parse_pt<false, true>(uint32_t(col) * 4u, bits, from, np, tabs0, tabsel, pt8);
      if (KM == 1 && mt)
        return count ? lj_guess_parse_mt<true, true>(uint32_t(col) * 4u, bits, from, tb_even, tb_odd, count)
                     : lj_guess_parse_mt<false, true>(uint32_t(col) * 4u, bits, from, tb_even, tb_odd);
#ifndef RSX_K0_SINGLE_SYMBOL
      if (hand)
        return count ? lj_guess_parse_pairs<true>(uint32_t(col) * 4u, bits, from, count)
                     : lj_guess_parse_pairs<false>(uint32_t(col) * 4u, bits, from);
#endif
      if (hand)
        return count ? lj_guess_parse_asm<true>(uint32_t(col) * 4u, bits, from, count)
                     : lj_guess_parse_asm<false>(uint32_t(col) * 4u, bits, from);
      return count ? lj_guess_parse<true>(lds_addr(&L.B[col]), lds_addr(lut8), bits, from, count)
                   : lj_guess_parse<false>(lds_addr(&L.B[col]), lds_addr(lut8), bits, from);
    };
    const uint32_t eb = L.ob[j];
    const bool exists = eb != 0u && !(j == 0 && lb == 0);
    uint32_t ea = 0, cnt = 0, grid = 0;
    bool constant = false;
    if (pt && zl != 0u)
      constant = lj_guess_constant_pt(L.B, j, zl, zc, zcum, np, exists, &ea, &cnt, &grid, eb);
    else if (zl >= 4u) // (shorter: more than 128 symbols in a slot, the multi-kernel pipeline's)
      constant = lj_guess_constant(L.B, j, zl, zc, exists, &ea, &cnt, &grid, zlb, eb);
    if (!constant && exists)
      ea = ((mt || pt) ? parse_from_0(j, eb, &cnt) : parse(j, eb, 0u, &cnt)) & smask;
    if (j == 0 && lb == 0)
      ea = S.start_bit & smask; // (the stream's first slot starts where the stream does)
    EA[j] = uint16_t(ea);
    lds_barrier();
    K0_STAMP(5);
    const uint32_t xa = j >= 1 ? uint32_t(EA[j - 1]) : 0u;
    uint32_t ebv = ea;
#ifdef RSX_K0_X1 // (diagnostic build: the B parse without its count -- wrong counts, K0's time only)
    if (!constant && exists && xa != 0u && gs >= 2u)
      ebv = parse(j, eb, xa, nullptr) & smask;
#else
    // (two tables: the parse from bit 0 stands for the B parse where the predecessor ends on
    // bit 0 -- unless it met a special length: then once more, looking them up)
    if (!constant && exists && gs >= 2u && (xa != 0u || ((mt || pt) && (cnt >> 31) != 0u)))
      ebv = parse(j, eb, xa, &cnt) & smask;
#endif
    EB[j] = uint16_t(ebv);
    uint32_t handed = 0;
    // Hand-over (K0_CHAIN): the state this workgroup's chain leaves its last slot in IS the
    // next workgroup's entry state, and after the B parses it is final in 98 % of the
    // workgroups -- so lane 255 puts it out now, and lane 0 takes the predecessor's instead of
    // its own estimate (the parse of ONE slot from bit 0: off in 1.5 % of the workgroups with
    // one table, in 12 % with two, where offset AND table have to fall into step): behind
    // the rounds, and if it differs the rounds run once more -- slot 1 is re-parsed from it like
    // any slot whose predecessor's exit moved.  Without
    // this every mis-estimated workgroup is a slow one in the single-pass kernel (its first
    // slots' guesses are off: re-decode rounds), and every workgroup behind a slow one in
    // flight waits for it, twice: 17 us of a workgroup's 44 with two tables.
    // (a predecessor whose rounds move its last exit after all puts the new one out at its
    // end; the successor's word then says "estimate != true entry", as before)
    if (K0_CHAIN) {
      const uint32_t tag = 0x8000u | (a.run_parity << 14);
      if (j == LJ_T - 1 && lb + 1 < S.n_blocks)
        __hip_atomic_store(&a.k0e[b + 1], tag | (ebv & smask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      // (asked for now, looked at behind the rounds: the predecessor is at the same point of
      // its life, give or take -- waiting for it here cost K0 27 %)
      if (j == 0 && lb > 0)
        handed = __hip_atomic_load(&a.k0e[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (15 bits of symbols, 0x7FFF = too many to say; bit 15: the parse met a special entry)
    auto pack_cnt = [](uint32_t c) -> uint16_t {
      const uint32_t n = c & 0x7FFFFFFFu;
      return uint16_t((n > 0x7FFFu ? 0x7FFFu : n) | ((c >> 31) << 15));
    };
    ECNT[j] = pack_cnt(cnt);
    lds_barrier(); // (every EA[j - 1] has been read: the array becomes EU)
    K0_STAMP(6);
    EU[j] = uint16_t(xa);
    // the guess for slot j + 1 (lane 255's goes to the next workgroup's slot 1): stored now,
    // behind the rounds' parses, and again by the rounds for the slots they move -- stores
    // at the very end of a workgroup are latency nothing hides
    const uint32_t g1 = S.first_subseq + lb * LJ_OWN; // record of this workgroup's slot 1
    const bool stored = j >= 1 && (j < LJ_T - 1 || lb + 1 < S.n_blocks);
    // (a slot the first round parses again is stored by that round: one writer at a time.
    // The guesses are hints -- the kernel checks every one against the exit before it --,
    // so a rare stale one costs a re-decode round there, never a pixel.)
    const bool first_listed =
        !constant && exists && j >= 1 && gs >= 3u && uint32_t(EB[j - 1]) != xa;
    if (stored && !first_listed)
      a.sub_start[g1 + uint32_t(j)] = uint16_t(ebv);
    // rounds: a slot is parsed again when its predecessor's last exit is not what its own
    // last parse started from; constant slots never are (their symbol grid is read from
    // their bits -- a parse from a wrong entry would only carry the error down the region)
    // (per round a list length and a "some exit moved" word of its own: nothing to reset
    // between rounds, two barriers a round -- what the three-slot chain took before)
    // (a lambda, run a second time after the hand-over below: as ONE loop nest -- "for pass" around
    // the rounds -- the compiler made the rounds 10 % of K0 slower for everybody, cfg 3 0.26 -> 0.29 ms)
    // (cmp: the bits of a state that must agree -- all of them, or the offset alone: the first
    // rounds of a table-per-phase stream, see "the phase pass" below)
    auto run_rounds = [&](uint32_t cmp) -> bool {
      for (uint32_t round = 0; round < max_rounds && gs >= 3u; ++round) {
        const uint32_t x = j >= 1 ? uint32_t(EB[j - 1]) : 0u;
        if (!constant && exists && j >= 1 && ((x ^ uint32_t(EU[j])) & cmp) != 0u)
          glist[atomicAdd(&rw[2 * round], 1u)] = uint16_t(j);
        lds_barrier();
        const uint32_t nth = rw[2 * round];
        if (nth == 0)
          return true;
        if (j < 64) {
          bool changed = false;
          for (uint32_t k = uint32_t(j); k < nth; k += 64u) {
            const int c = int(glist[k]);
            const uint32_t from = uint32_t(EB[c - 1]);
            uint32_t n = 0;
            const uint32_t e = parse(c, L.ob[c], from, &n) & smask;
            if (((e ^ uint32_t(EB[c])) & cmp) != 0u)
              changed = true;
            if ((round == 0u || e != uint32_t(EB[c])) && (c < LJ_T - 1 || lb + 1 < S.n_blocks))
              a.sub_start[g1 + uint32_t(c)] = uint16_t(e);
            EB[c] = uint16_t(e);
            EU[c] = uint16_t(from);
            ECNT[c] = pack_cnt(n);
          }
          if (changed)
            rw[2 * round + 1] = 1u;
        }
        lds_barrier();
        if (rw[2 * round + 1] == 0u) // (no exit moved: every successor's entry still stands)
          return true;
      }
      return false;
    };
    bool settled = run_rounds(smask);
    bool phase_from_counts = false; // (this workgroup's phases are the look-back's, below)
    if (KM == 2) {
      // The phase pass (streams with a table per phase).  Where the tables of the components
      // DIFFER a parse in the wrong phase reads wrong lengths and falls into step with the true
      // one in offset and phase alike, as with two tables: the rounds above settle.  Where they
      // AGREE -- on the short codes of smooth image regions they do, for channels of one image --
      // nothing in the bits says which phase a parse is in: its offsets fall into step as with
      // one table, its phase stays whatever its start assumed, and the fixed point of the chain
      // travels down the workgroup one slot a round.  But the phase of a symbol is its index mod
      // N, and in such a stretch the COUNTS are right whatever the phase: a workgroup whose
      // rounds did not settle gives every slot the phase a prefix sum of the counts gives it,
      // from the phase the workgroup starts in -- which is the prefix of the counts of ALL
      // workgroups in front of it: a decoupled look-back over 2-bit sums, every workgroup takes
      // part --, parses the slots whose last parse started in another phase ONCE more from the
      // right one, all at a time, and is done if no exit and no count moved.
      uint32_t* PX = reinterpret_cast<uint32_t*>(smem + lj_k0_ptx_off(a.pt_np));
      uint32_t* const kp_now = a.k0p + size_t(a.run_parity) * (a.n_blocks_plan + 1u);
      if (j == 0) // (the next run's words of this block: clean when it looks at them)
        a.k0p[size_t(a.run_parity ^ 1u) * (a.n_blocks_plan + 1u) + b] = 0u;
      // (a workgroup whose rounds settled IS in step with the true chain, phase and all -- a
      // wrong phase cannot be consistent over 255 slots parsed from bit 0 each --: the phase it
      // ends in goes out at once, and it looks at nobody.  Only the others walk back, to the
      // nearest such word: next door as a rule.)
      if (pt && settled && j == 0)
        __hip_atomic_store(&kp_now[b],
                           K0P_INC | K0P_AGG | ((uint32_t(EB[LJ_T - 1]) >> ST_PHASE_SHIFT) & 3u),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pt && !settled) {
        const uint32_t my_c = (j >= 1 && exists) ? (uint32_t(ECNT[j]) & 0x7FFFu) : 0u;
        uint32_t incl = my_c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t y = uint32_t(__shfl_up(int(incl), o, 64));
          if ((j & 63) >= o)
            incl += y;
        }
        if ((j & 63) == 63)
          PX[j >> 6] = incl;
        lds_barrier();
        const uint32_t w0 = PX[0], w1 = PX[1], w2 = PX[2], w3 = PX[3];
        const uint32_t total = w0 + w1 + w2 + w3;
        uint32_t before = incl - my_c + ((j >> 6) >= 1 ? w0 : 0u) + ((j >> 6) >= 2 ? w1 : 0u) +
                          ((j >> 6) >= 3 ? w2 : 0u);
        // the workgroup's own sum out, then the look-back (the first wavefront: 64 words a pass)
        if (j == 0)
          __hip_atomic_store(&kp_now[b], K0P_AGG | (total % np), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j < 64) {
          uint32_t acc = 0, ok = 0;
          if (lb == 0) {
            ok = 1; // (the stream's first symbol is phase 0)
          } else {
            int64_t pos = int64_t(lb) - 1;
            for (uint32_t spins = 0; spins < (1u << 14) && !ok; ++spins) {
              const int64_t idx = pos - j;
              uint32_t w = K0P_INC; // (in front of the stream: phase 0)
              if (idx >= 0)
                w = __hip_atomic_load(&kp_now[S.first_block + uint32_t(idx)], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
              const uint64_t m_inc = __ballot((w & K0P_INC) != 0u);
              const uint64_t m_any = __ballot((w & (K0P_INC | K0P_AGG)) != 0u);
              const int f = m_inc ? __builtin_ctzll(m_inc) : 64;
              const uint64_t need = f == 64 ? ~0ull : ((2ull << f) - 1ull);
              if ((m_any & need) != need) {
                __builtin_amdgcn_s_sleep(2);
                continue; // (a predecessor in the window has nothing out yet)
              }
              uint32_t v = (j <= f) ? (w & 3u) : 0u;
#pragma unroll
              for (int o = 32; o > 0; o >>= 1)
                v += uint32_t(__shfl_xor(int(v), o, 64));
              acc += v;
              if (f < 64)
                ok = 1;
              else
                pos -= 64;
            }
          }
          if (j == 0) {
            PX[4] = ok ? acc % np : ((uint32_t(EB[0]) >> ST_PHASE_SHIFT) & 3u);
            PX[5] = ok;
            PX[6] = 0u; // "the phase pass moved an exit or a count"
          }
        }
        lds_barrier();
        const uint32_t a_b = PX[4];
        if (j == 0) {
          __hip_atomic_store(&kp_now[b], K0P_INC | K0P_AGG | ((a_b + total) % np), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
          if (!PX[5] && !settled)
            nlist[2] = 1u; // (no prefix in time: the workgroup's own estimate, "uncertain")
#ifdef RSX_EXPERIMENT
          if (!PX[5])
            atomicOr(&a.results[s].stat_why, 0x100000u);
#endif
          if (!settled)
            EB[0] = uint16_t((uint32_t(EB[0]) & ST_OFF_MASK) | (a_b << ST_PHASE_SHIFT));
        }
        lds_barrier();
        // every slot's entry: its predecessor's exit offset | the phase the counts give it
        bool redo = false;
        uint32_t want_in = 0;
        const bool relabel = !settled; // (workgroup-uniform)
        phase_from_counts = relabel;
        if (relabel && j >= 1 && exists) {
          want_in = (uint32_t(EB[j - 1]) & ST_OFF_MASK) | (((a_b + before) % np) << ST_PHASE_SHIFT);
          redo = !constant && want_in != uint32_t(EU[j]);
          if (constant && want_in != grid) {
            nlist[2] = 1u; // (a constant slot's grid and phase are read from its bits)
#ifdef RSX_EXPERIMENT
            atomicOr(&a.results[s].stat_why, 0x400000u);
#endif
          }
        }
        lds_barrier(); // (every EB[j - 1] has been read)
        if (redo) {
          uint32_t n2 = 0;
          const uint32_t e2 = parse(j, eb, want_in, &n2) & smask;
          if (((e2 ^ uint32_t(EB[j])) & ST_OFF_MASK) != 0u || pack_cnt(n2) != ECNT[j])
            PX[6] = 1u;
          EB[j] = uint16_t(e2);
          EU[j] = uint16_t(want_in);
          ECNT[j] = pack_cnt(n2);
        } else if (relabel && j >= 1 && exists && !constant) {
          // (parsed from this very state already: its exit phase follows from its count)
          EB[j] = uint16_t((uint32_t(EB[j]) & ST_OFF_MASK) |
                           (((a_b + before + my_c) % np) << ST_PHASE_SHIFT));
        }
        lds_barrier();
        if (relabel && PX[6] == 0u)
          settled = true; // (every slot parsed from its predecessor's exit, phase and all)
        if (PX[6] != 0u) { // (the published sums may be off now: nothing downstream trusts them blindly)
          if (j < 2 * int(LJ_GUESS_ROUNDS_PT))
            rw[j] = 0u;
          lds_barrier();
          settled = run_rounds(smask);
          if (j == 0)
            nlist[2] = 1u;
#ifdef RSX_EXPERIMENT
          if (j == 0)
            atomicOr(&a.results[s].stat_why, 0x200000u);
#endif
        }
        // the guesses once more, phases and all (what the B parse and the rounds stored had
        // the phases of their time)
        if (relabel && stored)
          a.sub_start[g1 + uint32_t(j)] = EB[j];
      }
    }
    K0_STAMP(7);
#ifndef RSX_K0_NO_FINAL_HANDOVER
    // (the hand-over once more, now that the rounds have run: marked final)
    if (K0_CHAIN && j == LJ_T - 1 && lb + 1 < S.n_blocks)
      __hip_atomic_store(&a.k0e[b + 1],
                         0xA000u | (a.run_parity << 14) | (uint32_t(EB[LJ_T - 1]) & smask),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (K0_CHAIN) {
      // the hand-over: the predecessor's word (asked for before the rounds; once more if it
      // was not out yet -- it is by now as a rule: the predecessor started earlier)
      if (j == 0) {
        nlist[3] = 0u;
        if (lb > 0) {
          const uint32_t tag = 0x8000u | (a.run_parity << 14);
          uint32_t v = handed;
          for (uint32_t spins = 0; spins < (1u << 12) && (v & 0xC000u) != tag; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&a.k0e[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#ifndef RSX_K0_NO_FINAL_HANDOVER
          // (the predecessor's word from behind ITS rounds, if it comes within a few polls:
          // in 1.3 % of the workgroups the rounds move the last slot's exit, and a workgroup
          // that started from the older state is a slow one in the single-pass kernel)
          for (uint32_t spins = 0; spins < uint32_t(RSX_K0_FINAL_POLLS) &&
                                   (v & 0xE000u) != (tag | 0x2000u);
               ++spins) {
            const uint32_t w = __hip_atomic_load(&a.k0e[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w & 0xC000u) == tag)
              v = w;
            if ((v & 0xE000u) != (tag | 0x2000u))
              __builtin_amdgcn_s_sleep(1);
          }
#endif
          // (a table per phase: the phase of EB[0] is the look-back's, absolute; a predecessor's
          // word from BEFORE its own phase pass -- not marked final -- only has an offset to give)
          uint32_t vs = v & smask;
          if (phase_from_counts && (v & 0xE000u) != (tag | 0x2000u))
            vs = (vs & ST_OFF_MASK) | (uint32_t(EB[0]) & ~ST_OFF_MASK & smask);
          if ((v & 0xC000u) == tag && vs != uint32_t(EB[0])) {
            EB[0] = uint16_t(vs);
            nlist[3] = 1u;
          }
        }
      }
      lds_barrier();
      if (nlist[3] != 0u) { // the entry moved: the rounds once more (slot 1 is listed by the first)
        lds_barrier(); // (everybody has read the word)
        if (j >= 3 && j < 4 + 2 * int(LJ_GUESS_ROUNDS))
          nlist[j] = 0u;
        if (KM == 2 && j < 2 * int(LJ_GUESS_ROUNDS_PT))
          rw[j] = 0u;
        lds_barrier();
        settled = run_rounds(smask);
      }
    }
    ebv = uint32_t(EB[j]);
    const uint32_t cnt_word = uint32_t(ECNT[j]);
    cnt = cnt_word & 0x7FFFu;
    // what the count cannot vouch for: a chain that did not settle, a constant slot whose
    // symbol grid is not where its predecessor ends, symbols the 10-bit table only
    // approximates, more symbols than the word holds
    {
      const uint32_t x = j >= 1 ? uint32_t(EB[j - 1]) : 0u;
      if (j >= 1 && (!settled || (constant && exists && x != grid) || (cnt_word & 0x8000u) ||
                     cnt == 0x7FFFu))
        nlist[2] = 1u;
#ifdef RSX_EXPERIMENT
      // (why, for scripts/exp_mt_why.py: bits 16.. of the stream's reasons word)
      if (j >= 1) {
        const uint32_t why = (!settled ? 0x10000u : 0u) |
                             ((constant && exists && x != grid) ? 0x20000u : 0u) |
                             ((cnt_word & 0x8000u) ? 0x40000u : 0u) |
                             (cnt == 0x7FFFu ? 0x80000u : 0u);
        if (why)
          atomicOr(&a.results[s].stat_why, why);
      }
#endif
    }
#ifdef RSX_EXPERIMENT
    // (per-slot record for the single-pass kernel's cross-check, scripts/exp_mt_why.py)
    if (a.sub_sums && j >= 1)
      a.sub_sums[S.first_subseq + lb * LJ_OWN + uint32_t(j - 1)] =
          make_uint2(cnt | (constant ? 0x10000u : 0u) | (eb << 17),
                     ebv | ((j >= 1 ? uint32_t(EB[j - 1]) : 0u) << 8) | (grid << 16));
#endif
    // the LDS level of the single-pass launches: the symbols of this workgroup's slots
    // 1..255 (parsed from bit 0: an estimate) against what a level stages
    uint32_t c = j == 0 ? 0u : cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      c += uint32_t(__shfl_xor(int(c), o, 64));
    if ((j & 63) == 0)
      atomicAdd(est, c);
    lds_barrier();
    K0_STAMP(8);
    // The workgroup's word for the single-pass kernel's symbol base (LjArgs::k0w): symbols
    // of slots 1..255 (32 bits) | the exit of slot 0 the chain started from | bit 7: the
    // count is not to be trusted (16 bits) | the TRUE exit of slot 0 = the predecessor
    // workgroup's slot 255 under ITS chain, valid bit 15 (16 bits, written by that one).
    if (a.k0w) {
      uint8_t* w = reinterpret_cast<uint8_t*>(a.k0w + b);
      if (j == 0) {
        *reinterpret_cast<uint32_t*>(w) = *est;
        // (a state is offset | table bit or phase: 8 bits since round 6; "uncertain" is bit 8)
        *reinterpret_cast<uint16_t*>(w + 4) =
            uint16_t((uint32_t(EB[0]) & 0xFFu) | (nlist[2] ? 0x100u : 0u));
        if (lb == 0)
          *reinterpret_cast<uint16_t*>(w + 6) = uint16_t(0x8000u | (uint32_t(EB[0]) & 0xFFu));
      }
      if (j == LJ_T - 1 && lb + 1 < S.n_blocks)
        *reinterpret_cast<uint16_t*>(w + 8 + 6) = uint16_t(0x8000u | (ebv & 0xFFu));
    }
    if (K0_CHAIN && j == LJ_T - 1 && lb + 1 < S.n_blocks)
      __hip_atomic_store(&a.k0e[b + 1], 0xA000u | (a.run_parity << 14) | (ebv & smask),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j == 0) {
      const uint32_t need = *est + (*est >> 6) + 64u;
      uint32_t lv = 0;
      while (lv < 2u && need > a.fast_cap_lv[lv])
        ++lv;
      while (lv > 0u && a.fast_cap_lv[lv] == a.fast_cap_lv[lv - 1u])
        --lv; // (an absent level: the one below has the same LDS)
      if (lv)
        atomicMax(&a.fast_level[2u + a.run_parity], lv);
      // the lowest launched level that holds it, or the highest launched one
      uint32_t use = lv;
      while (use < 2u && !((a.fast_level_mask >> use) & 1u))
        ++use;
      while (use > 0u && !((a.fast_level_mask >> use) & 1u))
        --use;
      if (use)
        atomicMax(&a.fast_level[a.run_parity], use);
    }
    K0_STAMP(9);
  }
}

#undef K0_CHAIN

// inclusive scan of x over the wavefront
__device__ __forceinline__ uint32_t lj_wave_scan(uint32_t x, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o)
      x += y;
  }
  return x;
}

// ---------------------------------------------------------------------------
// Periodic data (stitch kernels only).  Slots with IDENTICAL content -- inside a
// constant region there are only a few different ones -- have identical transfer
// functions, and a slot's exit depends on nothing but its content and its entry
// state.  Up to PER_CLASSES classes of identical slots get a table: (exit, count, sums)
// for each of the 32 entry offsets, 256 decodes = one pass of the workgroup.  With the
// tables one lane walks the chain through the class slots with look-ups; what is left
// for the re-decode rounds are the few slots that are not in a class.
// ---------------------------------------------------------------------------
constexpr uint32_t LJ_SYNC_MAX_ROUNDS = 4, LJ_STITCH_MAX_ROUNDS = 32, LJ_STITCH_CLASS_ROUND = 3;
constexpr int PER_CLASSES = 8;

struct PeriodicEntry {
  uint16_t exit, count;
  uint32_t s0, s1;
};

struct PeriodicLds {
  uint32_t* hash;      // [LJ_T] content hash of every slot
  uint32_t* members;   // [LJ_T] slots whose representative is this one
  uint8_t* cls;        // [LJ_T] class of every slot (0xFF = none)
  uint32_t* rep_of;    // [PER_CLASSES] representative slot of a class
  PeriodicEntry* tbl;  // [PER_CLASSES * 32]
};
constexpr size_t lj_periodic_bytes() {
  return 2 * LJ_T * 4 + LJ_T + PER_CLASSES * 4 + PER_CLASSES * 32 * sizeof(PeriodicEntry) + 16;
}

__device__ __forceinline__ PeriodicLds carve_periodic(const Lds& L) {
  PeriodicLds p;
  p.hash = reinterpret_cast<uint32_t*>(sync_lds_end(L));
  p.members = p.hash + LJ_T;
  p.rep_of = p.members + LJ_T;
  p.tbl = reinterpret_cast<PeriodicEntry*>(p.rep_of + PER_CLASSES);
  p.cls = reinterpret_cast<uint8_t*>(p.tbl + PER_CLASSES * 32);
  return p;
}

// Classes of identical slots and their tables.  Called by the whole workgroup.
template <int NS, int BWK, typename TB>
__device__ __forceinline__ void lj_periodic_build(const Lds& L, const PeriodicLds& P,
                                                  const DecodeParams& dp, int j) {
  // content = the slot's dwords + its data-bit count
  uint32_t h = L.ob[j];
#pragma unroll
  for (int k = 0; k < BWK; ++k)
    h = (h ^ L.B[k * LJ_T + j]) * 0x9E3779B1u + (h >> 15);
  P.hash[j] = h;
  P.members[j] = 0;
  P.cls[j] = 0xFF;
  __syncthreads();
  // representative = the first slot (>= 1) with the same content
  uint32_t rep = uint32_t(j);
  for (uint32_t i = 1; i < uint32_t(LJ_T); ++i)
    if (P.hash[i] == h && i < rep)
      rep = i;
  if (rep != uint32_t(j)) {
    bool same = L.ob[rep] == L.ob[j];
    for (int k = 0; k < BWK; ++k)
      same = same && L.B[k * LJ_T + rep] == L.B[k * LJ_T + j];
    if (!same)
      rep = uint32_t(j); // (a hash collision: the slot stays on its own)
  }
  if (j >= 1)
    atomicAdd(&P.members[rep], 1u);
  __syncthreads();
  // the first PER_CLASSES representatives with at least 3 members become classes
  if (j == 0) {
    uint32_t nc = 0;
    for (uint32_t i = 1; i < uint32_t(LJ_T) && nc < uint32_t(PER_CLASSES); ++i)
      if (P.members[i] >= 3u) {
        P.rep_of[nc] = i;
        P.members[i] = 0x80000000u | nc;
        ++nc;
      }
    for (uint32_t c = nc; c < uint32_t(PER_CLASSES); ++c)
      P.rep_of[c] = 0;
  }
  __syncthreads();
  if (j >= 1 && (P.members[rep] & 0x80000000u))
    P.cls[j] = uint8_t(P.members[rep] & 0xFFu);
  // table entry t: class t / 32 decoded from entry offset t % 32
  const uint32_t c = uint32_t(j) >> 5, e0 = uint32_t(j) & 31u;
  const uint32_t r = P.rep_of[c];
  uint32_t e = ST_ERR, n = 0;
  uint2 sums = make_uint2(0, 0);
  lj_decode_span<false, NS, false, BWK, TB>(L, dp, int(r ? r : 1u), e0, L.ob[r ? r : 1u], e, n,
                                            &sums, r != 0);
  PeriodicEntry t;
  t.exit = uint16_t(e);
  t.count = uint16_t(n);
  t.s0 = sums.x;
  t.s1 = sums.y;
  P.tbl[j] = t;
  __syncthreads();
}

// One lane: follow the chain through the slots that belong to a class.
template <int NS>
__device__ __forceinline__ void lj_periodic_walk(const Lds& L, const PeriodicLds& P,
                                                 uint32_t true_start) {
  for (int q = 1; q < LJ_T; ++q) {
    const uint32_t want = q == 1 ? true_start : rec_st(L.rec[q - 1]);
    if (want == rec_su(L.rec[q]) || (want & ST_ERR) || !(L.ob[q] != 0 || q == 1))
      continue;
    const uint32_t c = P.cls[q];
    if (c == 0xFFu || (want & ST_OFF_MASK) > 31u)
      continue; // left to the re-decode rounds
    const PeriodicEntry t = P.tbl[c * 32u + (want & 31u)];
    L.rec[q] = rec_make(want, t.exit, t.count);
    sm_set<NS>(L, q, make_uint2(t.s0, t.s1));
  }
}

// ---------------------------------------------------------------------------
// K1 / K2: synchronisation.  NS = interleaved components of a fused-path stream
// (its difference sums are recorded), 0 = none.
// ---------------------------------------------------------------------------
// Which synchronisation instantiation takes a stream: the MULTI one also takes
// single-table streams that need the full 11-bit LUT (its states carry the component
// phase, which stays 0 for them).
__device__ __forceinline__ bool lj_sync_multi(const LjStreamDev& S) {
  return S.n_tables > 1 || S.sync_lut11 != 0;
}
template <bool MULTI, bool PAIR>
using SyncTable = std::conditional_t<(!MULTI && !PAIR), TabLds10, TabLds>;

template <bool STITCH, bool MULTI, bool PAIR, int NS>
__global__ __launch_bounds__(LJ_T) void lj_sync_kernel(LjArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x;
  const uint32_t s = a.block_stream[b];
  const LjStreamDev& S = a.streams[s];
  if (lj_sync_multi(S) != MULTI || (S.pair != 0) != PAIR || int(S.direct) != NS)
    return; // another instantiation handles this stream
  if (!lj_pipeline_takes(a, s, S))
    return; // the single-pass kernel's
  constexpr int BWK = PAIR ? LJ_BW_SYNC_PAIR : LJ_BW_SYNC;
  constexpr int N = NS ? NS : 1;
  using TB = SyncTable<MULTI, PAIR>;
  const Lds L = carve_sync(smem, BWK, NS, size_t(MULTI ? S.n_tables : 1u) * sizeof(TB));
  const uint32_t lb = b - S.first_block;
  const int j = threadIdx.x;

  uint32_t true_start = 0;
  bool unresolved = false;
  if (STITCH) {
    // a workgroup comes here when its assumed start differs from its predecessor's
    // recorded exit, or when it gave up on its own re-decode rounds (periodic data)
    unresolved = (a.block_flags[b] & 1u) != 0;
    true_start = lb == 0 ? uint32_t(S.start_bit) : a.block_exit[b - 1];
    if (true_start & ST_ERR)
      return; // chain broken by an error before this workgroup
    if (true_start == a.block_start[b] && !unresolved)
      return; // consistent already
  }

#ifdef RSX_EXPERIMENT
  if (STITCH && j == 0)
    atomicAdd(&a.results[s].stat_stitch, 1u);
#endif
  if (TabBits<TB>::value == LUT_BITS)
    lj_stage_tables(L, a, S);
  else
    lj_stage_tables10(L, a, S);
  lj_load_image<BWK, true>(L, a, b, j); // ends with a barrier (tables complete, too)
  const uint32_t own_bits = L.ob[j];
  DecodeParams dp = lj_params(S);
  if (MULTI && S.n_tables == 1)
    dp.period = 1; // (here for its 11-bit LUT: the phase does not influence the parse and
                   // must stay out of the state -- it would never self-synchronise)
  dp.long_codes = lj_long_codes<TB>(L, S.n_tables);
  const uint32_t gsub = S.first_subseq + lb * LJ_OWN + uint32_t(j - 1); // j >= 1

  // initial decode / initial records
  if (!STITCH) {
    uint32_t start = 0, e = 0, c = 0;
    uint2 sums = make_uint2(0, 0);
    // slot 0 of the first workgroup lies before the stream: its "exit" is the
    // known start state; every other slot decodes from its warm-up guess
    // (slot 0 of later workgroups from bit 0)
    const bool real_slot = !(lb == 0 && j == 0);
    const uint32_t guess = (LJ_ABLATE & 16u) ? 0u : lj_warmup<MULTI, PAIR, BWK, TB>(L, dp, j);
    if (j >= 2 || (j == 1 && lb > 0))
      start = guess;
    else if (j == 1)
      start = S.start_bit; // the stream's first symbol
    lj_decode_span<MULTI, NS, PAIR, BWK, TB>(L, dp, j, start, own_bits, e, c, &sums,
                                             real_slot && !(LJ_ABLATE & 4u));
    if (!real_slot) {
      e = S.start_bit;
      c = 0;
    }
    L.rec[j] = rec_make(start, e, c);
    sm_set<NS>(L, j, sums);
  } else {
    if (j == 0) {
      L.rec[0] = 0;
    } else {
      const uint32_t rec = a.sub_state[gsub];
      // (su = the state the slot was decoded FROM: a workgroup that gave up on its
      // rounds left records that are not a consistent chain yet)
      L.rec[j] = rec_make(a.sub_start[gsub], rec & ST_MASK, rec >> 16);
      if (NS)
        sm_set<NS>(L, j, a.sub_sums[gsub]);
    }
  }

#ifdef RSX_EXPERIMENT
  if (j == 0)
    L.misc[12] = 0;
#endif
  // Jacobi iteration with a dense work list: a slot whose recorded start state
  // differs from its predecessor's exit is re-decoded; the (few) such slots are
  // packed onto the first lanes so that a handful of stragglers do not cost a
  // whole-workgroup pass.
  const int first_chained = STITCH ? 2 : 1;
  uint32_t rounds = 0;
  bool gave_up = false, classes_ready = false;
  // the class tables of the stitch pass are per entry OFFSET: single-table streams only
  // (also the ones the multi-table instantiation takes for their 11-bit LUT: their phase stays 0)
  const bool classes_ok = STITCH && !PAIR && (!MULTI || S.n_tables == 1);
  const PeriodicLds PL = carve_periodic(L);
  while (true) {
    if (classes_ok && classes_ready) {
      // slots with identical content: their exits come from the class tables
      if (j == 0)
        lj_periodic_walk<NS>(L, PL, true_start);
      __syncthreads();
    }
    if (j == 0)
      L.misc[8] = 0;
    __syncthreads();
    const uint32_t my_su = rec_su(L.rec[j]);
    uint32_t want = my_su;
    if (STITCH && j == 1)
      want = true_start;
    else if (j >= first_chained)
      want = rec_st(L.rec[j - 1]);
    // Slots past the end of the data hold no symbol start: whatever state enters
    // them leaves them unchanged, and nothing after them is ever decoded (only a
    // suffix of a stream can be empty).  Chaining them would cost one round PER
    // SLOT -- up to 254 rounds in the last workgroup of every stream, which is
    // dispatched last and was the tail of the whole kernel.  (The stitch pass still
    // takes the entry state of an all-empty workgroup so that K3's chain check holds.)
    const bool chained = own_bits != 0u || (STITCH && j == 1);
    // An ERROR exit is not propagated either.  A slot decoded from a wrong guess can
    // run into an invalid code; passing that "exit" on made every later slot of the
    // workgroup re-decode into an error one round after the other, with the repair
    // following one slot behind -- up to 255 rounds (seen: 230 on a Hasselblad
    // frame).  Instead a slot whose predecessor currently ends in an error is left
    // alone: if the error is transient the predecessor is repaired and the slot is
    // compared again next round; if it is real, nothing after it is needed (the
    // decode kernel reports it from the failing slot's own record).
    if (want != my_su && chained && !(want & ST_ERR)) {
      const uint32_t k = atomicAdd(&L.misc[8], 1u);
      L.list[k] = uint16_t(j);
    }
    __syncthreads();
    const uint32_t n = L.misc[8];
    if (n == 0 || (LJ_ABLATE & 32u))
      break;
    // Constant image regions (blown highlights, masked borders, the padding of DNG
    // tiles) make the bit stream periodic: a mis-aligned parse can cycle for ever
    // without meeting the true one, the start of such a slot is only known from its
    // predecessor, and the rounds advance ONE slot each -- up to 255 rounds of one
    // active lane.  So the rounds are limited.  K1 gives up (the workgroup is flagged
    // and the stitch pass takes it); the stitch pass, which is launched for a handful
    // of workgroups and can afford the LDS, then builds the exits of the slots with
    // identical content for every entry state once (lj_periodic_build) and walks the
    // chain through them with table look-ups.
    ++rounds;
    if (rounds > (STITCH ? LJ_STITCH_MAX_ROUNDS : LJ_SYNC_MAX_ROUNDS)) {
      gave_up = true;
      break;
    }
    // (a round earlier for a workgroup that has given up before: it is chaining)
    if (classes_ok && !classes_ready &&
        rounds == (unresolved ? 2u : LJ_STITCH_CLASS_ROUND)) {
      lj_periodic_build<NS, BWK, TB>(L, PL, dp, j); // (workgroup-uniform; has barriers)
      classes_ready = true;
      continue;
    }
#ifdef RSX_EXPERIMENT
    if (j == 0) {
      atomicAdd(&a.results[s].stat_rounds, 1u);
      atomicAdd(&a.results[s].stat_redo, n);
      atomicMax(&a.results[s].pad2,
                ((++L.misc[12]) << 16) | (lb & 0x7FFFu) | (STITCH ? 0x8000u : 0u));
    }
#endif
    uint32_t idx = 0, w = 0, e = 0, c = 0;
    uint2 sums = make_uint2(0, 0);
    // only the waves that hold list entries do anything (wave-uniform test)
    if (uint32_t(j & ~63) < n) {
      const bool mine = uint32_t(j) < n;
      idx = mine ? uint32_t(L.list[j]) : 1u;
      w = (STITCH && idx == 1) ? true_start : rec_st(L.rec[idx - 1]);
      // (a plain re-decode of the whole slot.  Two ways to stop early where the new parse
      // meets the recorded one were measured and lost: a bitmap of the first 64 bits'
      // symbol starts (round 1) cannot carry the difference sums, and walking both
      // parses in lock step costs two steps per symbol while the slowest of the few
      // lanes of a round still runs to the end of its slot: 0.625 vs 0.596 ms per 8
      // cfg-3 frames.)
      lj_decode_span<MULTI, NS, PAIR, BWK, TB>(L, dp, int(idx), w, L.ob[idx], e, c, &sums,
                                               mine);
    }
    __syncthreads(); // every read of the records precedes the updates
    if (uint32_t(j) < n) {
      L.rec[idx] = rec_make(w, e, c);
      sm_set<NS>(L, int(idx), sums);
    }
  }

  const uint32_t my_rec = L.rec[j];
  const uint32_t my_count = j >= 1 ? rec_cn(my_rec) : 0u;
  const uint2 my_sums = (NS && j >= 1) ? sm_get<NS>(L, j) : make_uint2(0u, 0u);
  if (j >= 1) {
    a.sub_state[gsub] = rec_st(my_rec) | (my_count << 16);
    a.sub_start[gsub] = uint16_t(rec_su(my_rec));
    if (NS)
      a.sub_sums[gsub] = my_sums;
  }
  if (j == 1)
    a.block_start[b] = rec_su(my_rec);
  if (j == LJ_T - 1)
    a.block_exit[b] = rec_st(my_rec);
  // A workgroup of periodic data: its exit for EVERY entry state, by following the
  // chain with the class tables (slots outside a class must be entered the way they
  // were recorded).  lj_pchain_kernel strings these together so that a constant region
  // that spans many workgroups is settled by one more stitch pass, not one per workgroup.
  bool tf_written = false;
  if (classes_ok && classes_ready) {
    if (j < 32) {
      uint32_t state = uint32_t(j);
      for (int q = 1; q < LJ_T && state != 0xFFFFu; ++q) {
        if (!(L.ob[q] != 0 || q == 1) || (state & ST_ERR))
          continue;
        const uint32_t c = PL.cls[q];
        if (c != 0xFFu && (state & ST_OFF_MASK) <= 31u)
          state = PL.tbl[c * 32u + (state & 31u)].exit;
        else if (state == rec_su(L.rec[q]))
          state = rec_st(L.rec[q]);
        else
          state = 0xFFFFu; // unknown for this entry
      }
      a.block_tf[size_t(b) * 32 + j] = uint16_t(state);
    }
    tf_written = true;
  }
  if (j == 0) {
    a.block_flags[b] = (gave_up ? 1u : 0u) | (tf_written ? 2u : 0u);
    if (gave_up || tf_written)
      atomicOr(&a.results[s].flags, FL_PERIODIC);
  }
  // Per slot: the symbols before it inside the workgroup and (fused path) the
  // running sums P before it, by phases relative to the workgroup's first symbol (a
  // slot's own phases start at its first symbol: rotate by the number of symbols
  // before it).  Block totals: symbols, sums.
  const int lane = j & 63, wv = j >> 6;
  const uint32_t incl = lj_wave_scan(my_count, lane);
  if (lane == 63)
    L.misc[wv] = incl;
  __syncthreads();
  uint32_t before = incl - my_count;
  for (int w = 0; w < wv; ++w)
    before += L.misc[w];
  if (j == 0)
    a.block_sum[b] = L.misc[0] + L.misc[1] + L.misc[2] + L.misc[3];
  if (NS) {
    const uint2 r = lj_rot_fields<N>(my_sums, before & uint32_t(N - 1));
    uint2 pincl = r;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint2 y = make_uint2(__shfl_up(pincl.x, o, 64), __shfl_up(pincl.y, o, 64));
      if (lane >= o)
        pincl = pk_add2(pincl, y);
    }
    if (lane == 63) {
      L.misc[4 + 2 * wv] = pincl.x;
      L.misc[5 + 2 * wv] = pincl.y;
    }
    __syncthreads();
    uint2 pex = pk_sub2(pincl, r);
    for (int w = 0; w < wv; ++w)
      pex = pk_add2(pex, make_uint2(L.misc[4 + 2 * w], L.misc[5 + 2 * w]));
    if (j >= 1) {
      a.sub_first[gsub] = before;
      a.sub_psum[gsub] = pex;
    }
    if (j == LJ_T - 1)
      a.block_psum[b] = pk_add2(pex, r);
  }
}

// ---------------------------------------------------------------------------
// Fallback for streams that do not self-synchronise (constant image regions make
// the bit stream periodic, and a mis-aligned parse of a periodic stream can
// cycle forever without meeting the true one).  Propagating the true state
// workgroup by workgroup would take one stitch launch per 16 KB; instead every
// workgroup of such a stream computes its TRANSFER FUNCTION -- the exit state
// for each of the (at most 64 x period) possible entry states, one lane per
// entry state, each lane a plain sequential decode of the workgroup's 255
// slots -- and one lane per stream then chains the functions.  After that every
// workgroup knows its true entry state and a single stitch pass finishes the job.
// ---------------------------------------------------------------------------
constexpr int TF_ENTRIES = 512; // index = state & 0x1FF (offset | phase << 6)

// The un-stuffed image is read straight from global memory here (all lanes of
// a wavefront read the same dwords): without the LDS image the kernel is
// limited by wave slots, not LDS, and every workgroup of the plan is resident.
template <bool MULTI, bool PAIR = false>
__global__ __launch_bounds__(LJ_T) void lj_transfer_kernel(LjArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x;
  const uint32_t s = a.block_stream[b];
  const LjStreamDev& S = a.streams[s];
  if ((S.n_tables > 1) != MULTI || (S.pair != 0) != PAIR ||
      !(a.results[s].flags & FL_UNCONVERGED) || !lj_pipeline_takes(a, s, S))
    return;
  Lds L{};
  L.B = const_cast<uint32_t*>(
      reinterpret_cast<const uint32_t*>(a.unstuffed + size_t(b) * LJ_IMG_U4));
  const uint32_t* ob32 = L.B + LJ_BW * LJ_T; // the image keeps ob[] as dwords
  L.tabs = reinterpret_cast<TabLds*>(smem);
  const int j = threadIdx.x;
  lj_stage_tables(L, a, S);
  __syncthreads();
  const DecodeParams dp = lj_params(S);
  // lane -> entry state: offset 0..63 (0..31 | phase << 6 with several tables)
  const uint32_t off = MULTI ? uint32_t(j) & 31u : uint32_t(j) & 63u;
  const uint32_t phase = MULTI ? uint32_t(j) >> 5 : 0u;
  const bool enabled = !MULTI || phase < S.period;
  uint32_t state = off | (phase << ST_PHASE_SHIFT);
  for (int slot = 1; slot < LJ_T; ++slot) {
    uint32_t e = ST_ERR, c = 0;
    lj_decode_span<MULTI, 0, PAIR, 0>(L, dp, slot, state, ob32[slot], e, c, nullptr, enabled);
    state = e;
  }
  if (enabled)
    a.transfer[size_t(b) * TF_ENTRIES + (off | (phase << ST_PHASE_SHIFT))] =
        uint16_t(state & ST_MASK);
}

__global__ __launch_bounds__(64) void lj_chain_kernel(LjArgs a) {
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (threadIdx.x != 0 || !(a.results[s].flags & FL_UNCONVERGED) ||
      !lj_pipeline_takes(a, s, S))
    return;
  uint32_t state = S.start_bit;
  for (uint32_t lb = 0; lb < S.n_blocks; ++lb) {
    const uint32_t b = S.first_block + lb;
    state = (state & ST_ERR) ? ST_ERR
                             : uint32_t(a.transfer[size_t(b) * TF_ENTRIES + (state & 0x1FFu)]);
    a.block_exit[b] = state;
  }
}

// Chain of the workgroups of a stream that holds periodic data (FL_PERIODIC): the
// stitch pass has left, for the workgroups it took, the exit for every entry state
// (block_tf).  One wavefront per stream walks the workgroups in order, 64 at a time
// (records coalesced into registers, the tables into LDS), and rewrites the recorded
// exits of those workgroups with the ones their TRUE entry states lead to; the next
// stitch pass then re-converges every workgroup whose assumed start has moved.
__global__ __launch_bounds__(64) void lj_pchain_kernel(LjArgs a) {
  __shared__ uint16_t tf[64][32];
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  if (!(a.results[s].flags & FL_PERIODIC) || !lj_pipeline_takes(a, s, S))
    return;
  const int lane = threadIdx.x;
  const uint32_t fb = S.first_block, nb = S.n_blocks;
  uint32_t state = S.start_bit;
  for (uint32_t c0 = 0; c0 < nb; c0 += 64) {
    const uint32_t i = c0 + uint32_t(lane);
    uint32_t bs = 0, be = 0, fl = 0;
    if (i < nb) {
      bs = a.block_start[fb + i];
      be = a.block_exit[fb + i];
      fl = a.block_flags[fb + i];
      if (fl & 2u) {
        const uint4* src = reinterpret_cast<const uint4*>(a.block_tf + size_t(fb + i) * 32);
        uint4* dst = reinterpret_cast<uint4*>(tf[lane]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          dst[k] = src[k];
      }
    }
    __syncthreads();
    uint32_t mine = be;
    const uint32_t n_here = nb - c0 < 64u ? nb - c0 : 64u;
    if (__ballot((fl & 2u) != 0) == 0ull) {
      // no table among these 64 workgroups: whatever state enters one of them, the
      // recorded exit is all there is -- the chain leaves with the last one's
      if (!(state & ST_ERR))
        state = __shfl(be, int(n_here - 1), 64);
      __syncthreads();
      continue;
    }
    for (uint32_t k = 0; k < n_here; ++k) {
      const uint32_t bs_k = __shfl(bs, int(k), 64), be_k = __shfl(be, int(k), 64);
      const uint32_t fl_k = __shfl(fl, int(k), 64);
      uint32_t ex = be_k;
      if (state & ST_ERR) {
        ex = ST_ERR;
      } else if ((fl_k & 2u) && (state & ~31u) == 0u && tf[k][state & 31u] != 0xFFFFu) {
        ex = tf[k][state & 31u];
      } else if (state != bs_k) {
        ex = be_k; // unknown: the recorded exit is the best guess there is
      }
      if (uint32_t(lane) == k)
        mine = ex;
      state = ex;
    }
    if (i < nb && (fl & 2u) && !(mine & ST_ERR) && mine != be)
      a.block_exit[fb + i] = mine;
    __syncthreads();
  }
}

// run-time flavour of lj_rot_fields (the per-stream scan kernel is not
// instantiated per component count)
__device__ __forceinline__ uint2 lj_rot_fields_rt(uint2 v, uint32_t f, uint32_t n) {
  return n == 2 ? lj_rot_fields<2>(v, f) : (n == 4 ? lj_rot_fields<4>(v, f) : v);
}

// ---------------------------------------------------------------------------
// K3: per stream -- chain check; exclusive scans over the workgroups of the symbol
// counts, the dropped stuffing bytes and (fused path) the difference sums: the
// latter become P, the running sum of every component's differences over the
// whole stream, before each workgroup's first symbol.
// ---------------------------------------------------------------------------
__device__ void lj_consumed_body(const LjArgs& a, uint32_t s, int lane);
__device__ __forceinline__ uint32_t lj_zero_bytes(uint32_t d) { // 0x80 per zero byte, exact
  return ~(((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d | 0x7F7F7F7Fu);
}

// The scan of a single-pass stream's first pass, LJ_T * K workgroups at a time: a thread owns K
// consecutive workgroups and asks for everything the scan and its checks read of them AT ONCE.
// (The loop below reads its seven words under the conditions that need them -- i >= 1, i < nb,
// "delivered symbols behind it" --, which the compiler turns into as many memory round trips
// in a row: 4 us for every 256 workgroups of a stream, measured as 0.023 / 0.037 / 0.063 ms
// of kernel for streams of 680 / 1930 / 3200 workgroups.)  carry / dcarry: the symbols and
// dropped bytes in front of `base`; returns whether a check failed for this thread.
template <int K>
__device__ __forceinline__ bool lj_scan_first_pass(const LjArgs& a, uint32_t fb, uint32_t nb,
                                                   uint64_t needed, uint32_t base,
                                                   uint32_t* carry, uint32_t* dcarry,
                                                   uint32_t* wsum, uint32_t* dsum, int tid,
                                                   uint32_t cap_base_i, uint32_t cap_drop_i,
                                                   uint32_t* cap) {
  const uint32_t n_here = nb - base < uint32_t(LJ_T * K) ? nb - base : uint32_t(LJ_T * K);
  const uint32_t kk = (n_here + uint32_t(LJ_T) - 1u) / uint32_t(LJ_T); // (<= K)
  const uint32_t i0 = base + uint32_t(tid) * kk;
  uint32_t v[K], dv[K], b0[K], st[K], ex[K], fl[K];
#pragma unroll
  for (int q = 0; q < K; ++q) {
    // (every address valid, every load unconditional; what lies outside is dropped below)
    const uint32_t i = i0 + uint32_t(q);
    const uint32_t ic = (uint32_t(q) < kk && i < nb) ? i : nb - 1u;
    v[q] = a.block_sum[fb + ic];
    dv[q] = a.block_drops[fb + ic];
    b0[q] = a.block_base0[fb + ic];
    st[q] = a.block_start[fb + ic];
    ex[q] = a.block_exit[fb + (ic ? ic - 1u : 0u)];
    fl[q] = a.block_flags[fb + ic];
  }
  uint32_t run = 0, drun = 0;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const bool in = uint32_t(q) < kk && i0 + uint32_t(q) < nb;
    v[q] = in ? v[q] : 0u;
    dv[q] = in ? dv[q] : 0u;
    run += v[q];
    drun += dv[q];
  }
  // exclusive scan of the threads' totals
  uint32_t x = run, dx = drun;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    const uint32_t dy = __shfl_up(dx, o, 64);
    if ((tid & 63) >= o) {
      x += y;
      dx += dy;
    }
  }
  if ((tid & 63) == 63) {
    wsum[tid >> 6] = x;
    dsum[tid >> 6] = dx;
  }
  __syncthreads();
  uint32_t excl = *carry + x - run, dexcl = *dcarry + dx - drun;
  for (int w = 0; w < (tid >> 6); ++w) {
    excl += wsum[w];
    dexcl += dsum[w];
  }
  *carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  *dcarry += dsum[0] + dsum[1] + dsum[2] + dsum[3];
  bool bad = false;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const uint32_t i = i0 + uint32_t(q);
    if (uint32_t(q) < kk && i < nb) {
      a.block_base[fb + i] = excl;
      a.block_drop_base[fb + i] = dexcl;
      // (what the kernel's tail wants of the scan: kept in LDS instead of read back)
      if (i == cap_base_i)
        cap[0] = excl;
      if (i == cap_drop_i)
        cap[1] = dexcl;
      // (the checks of the loop in lj_scan_kernel, for fast_first: see there)
      const bool matters = uint64_t(excl) < needed;
      const bool link_broken = i >= 1u && st[q] != ex[q] && !(ex[q] & ST_ERR);
      bad = bad || ((fl[q] & 1u) != 0u && matters) || (link_broken && matters) ||
            (b0[q] != excl && (matters || uint64_t(b0[q]) < needed));
    }
    excl += v[q];
    dexcl += dv[q];
  }
  __syncthreads(); // (wsum / dsum are free again)
  return bad;
}

template <bool INV>
__global__ __launch_bounds__(LJ_T) void lj_scan_kernel(LjArgs a) {
  __shared__ uint32_t wsum[4], dsum[4];
  __shared__ uint2 psum[4];
  __shared__ uint32_t carry_s, dcarry_s;
  __shared__ uint2 pcarry_s;
  __shared__ uint32_t unconv_s;
  __shared__ uint32_t tail_s[16]; // (batched path: [0..3] part, [4..7] / [8..11] stuffing bytes, [12], [13] captures, [14] done)
#ifdef RSX_EXPERIMENT
  __shared__ uint32_t dbg_first_s;
  if (threadIdx.x == 0)
    dbg_first_s = 0xFFFFFFFFu;
#endif
  const uint32_t s = blockIdx.x;
  lj_fresh_scalars<INV>();
  const LjStreamDev& S = a.streams[s];
  if (!lj_bookkeeping_takes(a, s, S))
    return;
  const int tid = threadIdx.x;
  const uint32_t fb = S.first_block, nb = S.n_blocks;
  const uint32_t nd = S.direct;
  if (tid == 0) {
    carry_s = 0;
    dcarry_s = 0;
    pcarry_s = make_uint2(0, 0);
    unconv_s = 0;
  }
  __syncthreads();
  // (single-pass streams, first pass -- every run's, unless the kernel gave a stream up)
#ifndef RSX_EXPERIMENT // (experiment builds keep the loop: it records the first workgroup that fails a check)
  const bool batched = S.fast && a.pass == 0 && a.block_base0 != nullptr;
#else
  const bool batched = false;
#endif
  if (batched) {
    // Everything the tail of this kernel and the consumed-bytes rule (lj_consumed_body) read
    // that does NOT depend on the scan is asked for here, in front of it, by all 256 lanes:
    // the stream's result record, the symbol counts of the slots in front of the end of data,
    // the bytes of the workgroup region that holds the last symbol and the end of data (their
    // stuffing bytes are counted 16 bytes a lane, four pieces in flight).  Behind the scan the
    // tail is arithmetic.  (As it was, the tail was eight to ten memory round trips in a row
    // on one wavefront: 0.019 ms of kernel for a stream of 680 workgroups whose scan takes 3 us.)
    // (through a vector register: written by the kernels in front of this one in THIS run, and
    // nothing invalidates a CU's scalar cache between two kernels of a stream)
    uint32_t sv = s;
    asm volatile("" : "+v"(sv));
    const LjResult Rv = a.results[sv];
    const uint64_t needed = S.needed;
    const uint64_t dend = lj_data_end(S), in_bytes = S.in_bytes;
    const bool has_marker = Rv.marker_pos != 0xFFFFFFFFu && uint64_t(Rv.marker_pos) < in_bytes;
    const uint64_t M_av = uint64_t(Rv.marker_pos) < dend ? uint64_t(Rv.marker_pos) : dend;
    const uint64_t M_c = has_marker ? uint64_t(Rv.marker_pos) : dend;
    const uint64_t lbm_av = M_av / LJ_R; // (the workgroup region the end of data lies in)
    const uint64_t slot_phys = uint64_t(Rv.last_slot) * LJ_P;
    uint64_t lbs = slot_phys / LJ_R, lbm = M_c / LJ_R;
    lbs = lbs >= nb ? nb - 1 : lbs;
    lbm = lbm >= nb ? nb - 1 : lbm;
    // the rule's common case: the last symbol's slot and the end of data in one region
    const bool quick = a.fuse_consumed && !S.raw && !S.pair && lbs == lbm && slot_phys <= M_c &&
                       M_c <= (lbs + 1) * uint64_t(LJ_R) && nb != 0;
    uint32_t part = 0;
    if (lbm_av < nb) {
      const uint32_t js = uint32_t((M_av - lbm_av * LJ_R) / LJ_P);
      const uint32_t g0 = S.first_subseq + uint32_t(lbm_av) * LJ_OWN;
      if (uint32_t(tid) <= js && tid < LJ_OWN)
        part = a.sub_state[g0 + uint32_t(tid)] >> 16;
    }
    uint32_t drops_a = 0, drops_b = 0; // stuffing bytes in [region start, slot), [slot, end of data)
    if (quick) {
      const uint8_t* in = a.in_base + S.in_offset;
      const uint64_t r0 = lbs * uint64_t(LJ_R);
      uint4 v[4];
      uint32_t pv[4];
      bool whole[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t p0 = r0 + (uint64_t(u) * LJ_T + uint32_t(tid)) * 16u;
        whole[u] = p0 < M_c && p0 + 16 <= in_bytes;
        v[u] = make_uint4(0, 0, 0, 0);
        pv[u] = 0;
        if (whole[u]) {
          __builtin_memcpy(&v[u], in + p0, 16);
          pv[u] = p0 > 0 ? in[p0 - 1] : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t p0 = r0 + (uint64_t(u) * LJ_T + uint32_t(tid)) * 16u;
        if (p0 >= M_c)
          continue;
        uint32_t n = 0;
        if (whole[u]) {
          const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          uint32_t prev = pv[u];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // (bytes at and behind the end of data do not count: 0x80 per byte in front of it)
            const uint64_t q = p0 + 4u * uint32_t(k);
            const uint32_t live = q + 4 <= M_c ? 0x80808080u
                                               : (q >= M_c ? 0u : (0x80808080u >> (8u * uint32_t(q + 4 - M_c))));
            const uint32_t z = lj_zero_bytes(d[k]), f = lj_zero_bytes(~d[k]);
            n += uint32_t(__builtin_popcount(z & live & ((f << 8) | (prev == 0xFFu ? 0x80u : 0u))));
            prev = d[k] >> 24;
          }
        } else { // (the last bytes of the buffer)
          uint32_t prev = p0 > 0 ? in[p0 - 1] : 0u;
          for (uint64_t q = p0; q < M_c && q < in_bytes; ++q) {
            const uint32_t c = in[q];
            n += (c == 0u && prev == 0xFFu) ? 1u : 0u;
            prev = c;
          }
        }
        // (slots are 64 bytes: a 16-byte piece lies on one side of the slot's start)
        if (p0 < slot_phys)
          drops_a += n;
        else
          drops_b += n;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += uint32_t(__shfl_xor(int(part), o, 64));
      drops_a += uint32_t(__shfl_xor(int(drops_a), o, 64));
      drops_b += uint32_t(__shfl_xor(int(drops_b), o, 64));
    }
    if ((tid & 63) == 0) {
      tail_s[tid >> 6] = part;
      tail_s[4 + (tid >> 6)] = drops_a;
      tail_s[8 + (tid >> 6)] = drops_b;
    }
    // the scan
    uint32_t carry = 0, dcarry = 0;
    bool bad = false;
    for (uint32_t base = 0; base < nb;) {
      if (nb - base <= uint32_t(LJ_T) * 4u) {
        bad = lj_scan_first_pass<4>(a, fb, nb, needed, base, &carry, &dcarry, wsum, dsum, tid,
                                    uint32_t(lbm_av), uint32_t(lbs), tail_s + 12) || bad;
        base += uint32_t(LJ_T) * 4u;
      } else {
        bad = lj_scan_first_pass<16>(a, fb, nb, needed, base, &carry, &dcarry, wsum, dsum, tid,
                                     uint32_t(lbm_av), uint32_t(lbs), tail_s + 12) || bad;
        base += uint32_t(LJ_T) * 16u;
      }
    }
    if (bad)
      unconv_s = 1;
    __syncthreads();
    // the tail (what follows the loop below for the other streams), on lane 0
    bool consumed_done = false;
    if (tid == 0) {
      LjResult& R = a.results[s];
      const uint32_t avail = lbm_av >= nb ? carry
                                          : tail_s[12] + tail_s[0] + tail_s[1] + tail_s[2] + tail_s[3];
      R.avail_lo = avail;
      uint32_t flags = Rv.flags & ~(FL_UNCONVERGED | FL_NEED_LEGACY | FL_PERIODIC);
      if (unconv_s)
        flags |= FL_UNCONVERGED;
      if (uint64_t(avail) < needed) // (fused path or not: symbols past the data are the second pass's)
        flags |= FL_NEED_LEGACY;
      if (flags & (FL_UNCONVERGED | FL_NEED_LEGACY))
        flags = (flags & ~(FL_UNCONVERGED | FL_NEED_LEGACY)) | FL_SLOW;
      R.flags = flags;
      // K7's rule (lj_consumed_body) where it is arithmetic: the last refill touched the
      // marker, or ran off the end of the buffer
      if (quick && Rv.status == 0 && uint64_t(avail) >= needed && !Rv.tail_used) {
        const uint64_t drops_slot = uint64_t(tail_s[13]) + tail_s[4] + tail_s[5] + tail_s[6] + tail_s[7];
        const uint64_t c = (slot_phys - drops_slot) * 8 + Rv.last_pos;
        const uint64_t K = (c + 31) / 32 + 1;
        const uint64_t D = M_c - (drops_slot + tail_s[8] + tail_s[9] + tail_s[10] + tail_s[11]);
        if (4 * K > D) {
          R.consumed = uint32_t(has_marker ? M_c : in_bytes + (4 * K - D));
          consumed_done = true;
        }
      }
      // (a stream that is not to be looked at: status set, or symbols missing)
      if (a.fuse_consumed && (Rv.status != 0 || uint64_t(avail) < needed))
        consumed_done = true;
      tail_s[14] = consumed_done ? 1u : 0u;
    }
    if (a.fuse_consumed) {
      __threadfence_block();
      __syncthreads();
      if (tail_s[14] == 0u && tid < 64)
        lj_consumed_body(a, s, tid);
    }
    return;
  }
  for (uint32_t base = 0; base < nb; base += LJ_T) {
    const uint32_t i = base + tid;
    const uint32_t v = i < nb ? a.block_sum[fb + i] : 0u;
    const uint32_t dv = i < nb ? a.block_drops[fb + i] : 0u;
    // (a workgroup after a real error keeps whatever entry state it has: see K1)
    // (single-pass streams, first pass: judged below, where the workgroup's first symbol is
    // known -- what lies behind the last delivered symbol is nobody's business)
    const bool fast_first = S.fast && a.pass == 0;
    const bool link_broken = i >= 1 && i < nb &&
                             a.block_start[fb + i] != a.block_exit[fb + i - 1] &&
                             !(a.block_exit[fb + i - 1] & ST_ERR);

    // inclusive wave scans
    uint32_t x = v, dx = dv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o, 64);
      const uint32_t dy = __shfl_up(dx, o, 64);
      if ((tid & 63) >= o) {
        x += y;
        dx += dy;
      }
    }
    if ((tid & 63) == 63) {
      wsum[tid >> 6] = x;
      dsum[tid >> 6] = dx;
    }
    __syncthreads();
    uint32_t woff = 0, dwoff = 0;
    for (int w = 0; w < (tid >> 6); ++w) {
      woff += wsum[w];
      dwoff += dsum[w];
    }
    const uint32_t excl = carry_s + woff + x - v;
    const uint32_t dexcl = dcarry_s + dwoff + dx - dv;
    if (i < nb) {
      a.block_base[fb + i] = excl;
      a.block_drop_base[fb + i] = dexcl;
    }
    // A workgroup that is not settled -- its entry state is not its predecessor's exit, or
    // it gave up on its re-decode rounds -- keeps the stream unconverged only if a
    // DELIVERED symbol lies in or behind it.  (Found by the two-table fuzz of round 4: the
    // zeros behind the end-of-image marker end in an error state under some tables; the
    // workgroup behind that never runs again -- "chain broken by an error before this
    // workgroup" --, its gave-up flag stayed, and the host iterated to its limit and
    // reported a device error for a stream the reference decodes.  The counts in front of
    // the first unsettled workgroup are final, so its `excl` is.)
    const bool matters = uint64_t(excl) < S.needed;
    if (link_broken && !fast_first && matters)
      unconv_s = 1;
    if (i < nb && (a.block_flags[fb + i] & 1u) != 0 && matters)
      unconv_s = 1; // a workgroup that gave up on its re-decode rounds
    // The single-pass kernel took the index of a workgroup's first symbol from K0's counts
    // (+ the corrections of the workgroups K0 had flagged).  Here are the counts of its own
    // decodes: a workgroup that delivered symbols from another base than their sum has put
    // them in the wrong place (a count K0 got wrong without knowing: the multi-kernel
    // pipeline redoes the stream).  Workgroups behind the last delivered symbol do not matter.
    // (and the entry state it decoded from -- K0's chain of the workgroup before -- is what
    // that workgroup's decode arrived at, wherever delivered symbols follow)
    if (fast_first && link_broken && uint64_t(excl) < S.needed)
      unconv_s = 1;
    if (fast_first && a.block_base0 && i < nb && a.block_base0[fb + i] != excl &&
        (uint64_t(excl) < S.needed || uint64_t(a.block_base0[fb + i]) < S.needed)) {
      unconv_s = 1;
#ifdef RSX_EXPERIMENT
      if (atomicMin(&dbg_first_s, i) > i) { // (statistics: the first workgroup off its base)
        if (a.results[s].pad3[0] == 0u) {
          a.results[s].pad3[0] = 0x40000000u | i;
          a.results[s].pad3[1] = a.block_base0[fb + i];
          a.results[s].pad3[2] = excl;
        }
      }
#endif
    }
#ifdef RSX_EXPERIMENT
    if (fast_first && link_broken && uint64_t(excl) < S.needed &&
        atomicMin(&dbg_first_s, i) > i) {
      a.results[s].pad3[0] = 0x20000000u | i;
      a.results[s].pad3[1] = a.block_start[fb + i];
      a.results[s].pad3[2] = a.block_exit[fb + i - 1];
    }
    if (S.fast && a.pass == 0 && i < nb && (a.block_flags[fb + i] & 1u) != 0 &&
        atomicMin(&dbg_first_s, i) > i)
      a.results[s].pad3[0] = 0x10000000u | i;
#endif
    // (single-pass streams carry the predictor state through their own look-back: P before
    // each workgroup is the multi-kernel pipeline's, computed when it takes the stream over)
    if (nd && !fast_first) {
      // the workgroup's sums are kept by phases relative to its first symbol,
      // whose index is now known
      const uint2 pv = i < nb ? lj_rot_fields_rt(a.block_psum[fb + i], excl % nd, nd)
                              : make_uint2(0u, 0u);
      uint2 px = pv;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint2 py = make_uint2(__shfl_up(px.x, o, 64), __shfl_up(px.y, o, 64));
        if ((tid & 63) >= o)
          px = pk_add2(px, py);
      }
      if ((tid & 63) == 63)
        psum[tid >> 6] = px;
      __syncthreads();
      uint2 pex = pk_add2(pcarry_s, pk_sub2(px, pv));
      for (int w = 0; w < (tid >> 6); ++w)
        pex = pk_add2(pex, psum[w]);
      if (i < nb)
        a.block_pbase[fb + i] = pex;
      __syncthreads();
      if (tid == LJ_T - 1)
        pcarry_s = pk_add2(pex, pv);
    }
    __syncthreads();
    if (tid == LJ_T - 1) {
      carry_s = excl + v;
      dcarry_s = dexcl + dv;
    }
    __syncthreads();
  }
  // symbols that start before the end of data M = min(marker, in_bytes): those in front of
  // M's workgroup + those of its slots up to M's (a lane per slot; one lane walking them
  // was up to 255 load latencies in a row)
  {
    LjResult& R = a.results[s];
    uint64_t M = R.marker_pos;
    if (M > lj_data_end(S))
      M = lj_data_end(S);
    const uint64_t lbm = M / LJ_R;
    uint32_t part = 0;
    if (lbm < nb) {
      const uint32_t js = uint32_t((M - lbm * LJ_R) / LJ_P);
      const uint32_t g0 = S.first_subseq + uint32_t(lbm) * LJ_OWN;
      if (uint32_t(tid) <= js && tid < LJ_OWN)
        part = a.sub_state[g0 + uint32_t(tid)] >> 16;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      part += uint32_t(__shfl_xor(int(part), o, 64));
    if ((tid & 63) == 0)
      wsum[tid >> 6] = part;
  }
  __syncthreads();
  if (tid == 0) {
    LjResult& R = a.results[s];
    uint64_t M = R.marker_pos;
    if (M > lj_data_end(S))
      M = lj_data_end(S);
    uint32_t avail;
    const uint64_t lbm = M / LJ_R;
    if (lbm >= nb)
      avail = carry_s;
    else
      avail = a.block_base[fb + lbm] + wsum[0] + wsum[1] + wsum[2] + wsum[3];
    R.avail_lo = avail;
    uint32_t flags = R.flags & ~(FL_UNCONVERGED | FL_NEED_LEGACY | FL_PERIODIC);
    if (unconv_s)
      flags |= FL_UNCONVERGED;
    // symbols past the end of the data: the reference's end-of-stream semantics
    // live in the legacy tail kernel
    // (... and of the single-pass kernel's streams that are not the fused path's -- three
    // components, the streams that leave differences --: their second pass is the legacy route)
    if ((nd || (S.fast && a.pass == 0)) && uint64_t(avail) < S.needed)
      flags |= FL_NEED_LEGACY;
    // a single-pass stream with a broken chain or symbols past its data: the
    // multi-kernel pipeline redoes it (and finds out again, from its own records)
    if (S.fast && a.pass == 0 && (flags & (FL_UNCONVERGED | FL_NEED_LEGACY)))
      flags = (flags & ~(FL_UNCONVERGED | FL_NEED_LEGACY)) | FL_SLOW;
    R.flags = flags;
  }
  // Plans whose streams all take the single-pass kernel have nothing between this kernel
  // and lj_consumed_kernel: its work is done here, by the first wavefront (a launch less:
  // 10 us of a single frame's 235).
  if (a.fuse_consumed) {
    __threadfence_block();
    __syncthreads();
    if (tid < 64)
      lj_consumed_body(a, s, tid);
  }
}

// ---------------------------------------------------------------------------
// K4 (legacy path): final decode -> stream-ordered int16 differences.  Streams of
// the fused path come here only when they are damaged (FL_NEED_LEGACY).
// ---------------------------------------------------------------------------
// LAS: the stream's table holds Nikon "lossy after split" values (its own
// instantiation so that the JPEG hot loop carries no extra branch)
#ifndef RSX_K4_BURST
#define RSX_K4_BURST 4
#endif
constexpr int K4_BURST = RSX_K4_BURST; // groups of 8 differences per store burst

template <bool MULTI, bool LAS = false>
__global__ __launch_bounds__(LJ_T) void lj_decode_kernel(LjArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x;
  const uint32_t s = a.block_stream[b];
  const LjStreamDev& S = a.streams[s];
  if ((S.n_tables > 1) != MULTI || (S.las != 0) != LAS || S.pair)
    return;
  if (!lj_legacy_takes(a, s, S))
    return; // decoded by lj_decode_direct_kernel
  const Lds L = carve(smem, LJ_BW_DEC);
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
  lj_load_image<LJ_BW_DEC>(L, a, b, j); // ends with a barrier
  DecodeParams dp = lj_params(S);
  dp.long_codes = lj_long_codes(L, S.n_tables);

  const uint32_t gsub = S.first_subseq + lb * LJ_OWN + uint32_t(j - 1);
  uint32_t my_start = 0, my_count = 0, my_exit = 0;
  if (j >= 1) {
    const uint32_t rec = a.sub_state[gsub];
    my_count = rec >> 16;
    my_exit = rec & ST_MASK;
    my_start = (j == 1) ? a.block_start[b] : (a.sub_state[gsub - 1] & ST_MASK);
  }
  // exclusive scan of counts inside the workgroup
  uint32_t x = my_count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if ((j & 63) >= o)
      x += y;
  }
  if ((j & 63) == 63)
    L.misc[j >> 6] = x;
  __syncthreads(); // also: tables complete
  uint32_t woff = 0;
  for (int w = 0; w < (j >> 6); ++w)
    woff += L.misc[w];
  const uint64_t first = uint64_t(base) + woff + x - my_count; // first symbol of this lane

  // a bad Huffman code inside the delivered range is a real error
  // (PrefixCodeLookupDecoder.h:152-155)
  if (j >= 1 && (my_exit & ST_ERR) && first + my_count < needed &&
      int64_t(lb) * LJ_R + int64_t(j - 1) * LJ_P < int64_t(M))
    atomicCAS(&a.results[s].status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));

  // Every lane streams its own symbols to the stream-ordered scratch: 8
  // differences are packed in registers and leave as one 16-byte store (the
  // lane's output range starts on an arbitrary 2-byte boundary; gfx950 global
  // stores take unaligned addresses).  No LDS staging, no barriers.  The loop
  // is wave-uniform: every lane runs max-over-the-wave groups, lanes that are
  // past their last symbol stop advancing and their stores are masked.
  uint32_t remaining = (my_start & ST_ERR) ? 0u : my_count;
  if (first >= needed)
    remaining = 0;
  else if (first + remaining > needed)
    remaining = uint32_t(needed - first);
  int16_t* __restrict__ out = a.diffs + S.diff_offset + first;

  uint32_t wmax = remaining;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    wmax = max(wmax, uint32_t(__shfl_xor(wmax, o, 64)));
  const uint32_t n_groups = (wmax + 7) >> 3;

  uint32_t phase = (my_start >> ST_PHASE_SHIFT) & 7u;
  BitReader<LJ_BW_DEC> r;
  r.open(L.B, j, my_start & ST_OFF_MASK);
  uint32_t tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0; // the lane's last, partial group
  // A lane's 16-byte stores land on an arbitrary 2-byte boundary of a region it
  // shares cache lines with its neighbours'.  Issued one per group, every line
  // stayed partially written for the length of the loop, ~32 MB of such lines
  // device-wide against 32 MB of L2: they were evicted half-filled and the kernel
  // wrote 2.8x its output to HBM (PMC WRITE_SIZE, profiles/r01/ljpeg_traffic*.json).
  // So four groups are decoded into registers and leave as one 64-byte burst.
  auto decode_group = [&](uint32_t g, uint32_t (&p)[4]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool live = 8 * g + q < remaining;
      const uint32_t w = r.head();
      const uint32_t e = lj_entry(w, lj_table<MULTI>(L, dp, phase), live, dp.long_codes);
      r.advance(L.B, j, live ? (e >> 10) : 0u);
      if (MULTI)
        phase = live ? ((phase + 1 == dp.period) ? 0u : phase + 1) : phase;
      const uint32_t cl = e & 31u, ssss = (e >> 5) & 31u;
      uint32_t diff;
      if (LAS) { // NikonLASDecompressor::decodeDifference (.cpp:366-376)
        const uint32_t nbits = (e >> 10) - cl, shl = ssss - nbits;
        const uint32_t v = uint32_t((uint64_t(w << cl) << nbits) >> 32);
        int d = int((((v << 1) + 1u) << shl) >> 1);
        if ((d & (1 << ((ssss - 1u) & 31u))) == 0)
          d -= (1 << ssss) - (shl ? 0 : 1);
        diff = ssss == 16u ? 0x8000u : uint32_t(d);
        diff &= 0xFFFFu;
      } else {
        diff = lj_extend(w, e);
      }
      if (q & 1)
        p[q >> 1] |= diff << 16;
      else
        p[q >> 1] = diff;
    }
  };
  for (uint32_t g0 = 0; g0 < n_groups; g0 += K4_BURST) {
    uint32_t pv[K4_BURST][4];
#pragma unroll
    for (int u = 0; u < K4_BURST; ++u)
      if (g0 + u < n_groups) // wave-uniform
        decode_group(g0 + u, pv[u]);
#pragma unroll
    for (int u = 0; u < K4_BURST; ++u) {
      const uint32_t g = g0 + u;
      if (g >= n_groups)
        break;
      const uint32_t(&p)[4] = pv[u];
      if (8 * g + 8 <= remaining) {
        const uint4 v = make_uint4(p[0], p[1], p[2], p[3]);
        __builtin_memcpy(out + 8 * g, &v, 16);
      } else if (8 * g < remaining) {
        tp0 = p[0];
        tp1 = p[1];
        tp2 = p[2];
        tp3 = p[3];
      }
    }
  }
  // partial last group: 2-byte stores
  {
    const uint32_t full = remaining & ~7u, tail = remaining & 7u;
    for (uint32_t t = 0; t < tail; ++t) {
      const uint32_t word = (t >> 1) == 0 ? tp0 : ((t >> 1) == 1 ? tp1 : ((t >> 1) == 2 ? tp2 : tp3));
      out[full + t] = int16_t((t & 1) ? (word >> 16) : (word & 0xFFFFu));
    }
  }
  // K7 needs the bit position at which the reference's last symbol starts:
  // exactly one lane of the whole stream owns it and walks there again
  if (needed >= 1 && needed - 1 >= first && needed - 1 < first + remaining) {
    const uint32_t target = uint32_t(needed - 1 - first);
    uint32_t p2 = my_start & ST_OFF_MASK, ph2 = (my_start >> ST_PHASE_SHIFT) & 7u;
    for (uint32_t t = 0; t < target; ++t) {
      const uint32_t w = lj_window(L.B, j, p2);
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

// ---------------------------------------------------------------------------
// K4 for HasselbladDecompressor streams (HasselbladDecompressor.cpp:71-100): a
// step is a pair [len1 code][len2 code][len1 bits][len2 bits] and yields two
// differences; getBits() (.cpp:60-69) sign-extends like JPEG except that the
// all-ones 16-bit field means -32768.  Same skeleton as lj_decode_kernel: 4 pairs
// = 8 differences = one 16-byte store.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lj_hb_diff(uint32_t w, uint32_t ssss) {
  const uint32_t v = __builtin_amdgcn_ubfe(w, 32u - ssss, ssss);
  uint32_t d = (v >> ((ssss - 1u) & 31u)) ? v : v + 1u - (1u << ssss);
  d = (ssss == 16u && v == 0xFFFFu) ? 0x8000u : d;
  return d & 0xFFFFu;
}

__global__ __launch_bounds__(LJ_T) void lj_decode_pair_kernel(LjArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t b = blockIdx.x;
  const uint32_t s = a.block_stream[b];
  const LjStreamDev& S = a.streams[s];
  if (!S.pair || !lj_pipeline_takes(a, s, S))
    return;
  const Lds L = carve(smem);
  const uint32_t lb = b - S.first_block;
  const int j = threadIdx.x;
  const uint64_t needed = S.needed; // pairs
  const uint32_t base = a.block_base[b];
  const uint32_t sum = a.block_sum[b];
  if (base >= needed || sum == 0)
    return;
  if (uint64_t(lb) * LJ_R > lj_data_end(S))
    return;
  lj_stage_tables(L, a, S);
  lj_load_image(L, a, b, j); // ends with a barrier
  const DecodeParams dp = lj_params(S);
  const TabLds& tb = L.tabs[0];

  const uint32_t gsub = S.first_subseq + lb * LJ_OWN + uint32_t(j - 1);
  uint32_t my_start = 0, my_count = 0, my_exit = 0;
  if (j >= 1) {
    const uint32_t rec = a.sub_state[gsub];
    my_count = rec >> 16;
    my_exit = rec & ST_MASK;
    my_start = (j == 1) ? a.block_start[b] : (a.sub_state[gsub - 1] & ST_MASK);
  }
  uint32_t x = my_count;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if ((j & 63) >= o)
      x += y;
  }
  if ((j & 63) == 63)
    L.misc[j >> 6] = x;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < (j >> 6); ++w)
    woff += L.misc[w];
  const uint64_t first = uint64_t(base) + woff + x - my_count; // first pair of this lane
  if (j >= 1 && (my_exit & ST_ERR) && first + my_count < needed &&
      int64_t(lb) * LJ_R + int64_t(j - 1) * LJ_P < int64_t(lj_data_end(S)))
    atomicCAS(&a.results[s].status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));

  uint32_t remaining = (my_start & ST_ERR) ? 0u : my_count;
  if (first >= needed)
    remaining = 0;
  else if (first + remaining > needed)
    remaining = uint32_t(needed - first);
  uint32_t* __restrict__ out =
      reinterpret_cast<uint32_t*>(a.diffs + S.diff_offset) + first; // one dword per pair

  uint32_t wmax = remaining;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    wmax = max(wmax, uint32_t(__shfl_xor(wmax, o, 64)));
  const uint32_t n_groups = (wmax + 3) >> 2;

  uint32_t pos = my_start & ST_OFF_MASK;
  uint32_t tp[4] = {0, 0, 0, 0};
  for (uint32_t g = 0; g < n_groups; ++g) {
    uint32_t p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool live = 4 * g + q < remaining;
      const uint32_t e1 = lj_entry(lj_window(L.B, j, pos), tb, live);
      const uint32_t p1 = pos + (e1 & 31u);
      const uint32_t e2 = lj_entry(lj_window(L.B, j, p1), tb, live && e1 != 0u);
      const uint32_t p2 = p1 + (e2 & 31u);
      const uint32_t s1 = (e1 >> 5) & 31u, s2 = (e2 >> 5) & 31u;
      const uint32_t d1 = lj_hb_diff(lj_window(L.B, j, p2), s1);
      const uint32_t d2 = lj_hb_diff(lj_window(L.B, j, p2 + s1), s2);
      p[q] = d1 | (d2 << 16);
      pos = (live && e1 != 0u && e2 != 0u) ? p2 + s1 + s2 : pos;
    }
    if (4 * g + 4 <= remaining) {
      const uint4 v = make_uint4(p[0], p[1], p[2], p[3]);
      __builtin_memcpy(out + 4 * g, &v, 16);
    } else if (4 * g < remaining) {
      tp[0] = p[0]; tp[1] = p[1]; tp[2] = p[2]; tp[3] = p[3];
    }
  }
  {
    const uint32_t full = remaining & ~3u, tail = remaining & 3u;
    for (uint32_t t = 0; t < tail; ++t)
      out[full + t] = t == 0 ? tp[0] : (t == 1 ? tp[1] : tp[2]);
  }
  // K7 needs the bit position at which the last pair starts
  if (needed >= 1 && needed - 1 >= first && needed - 1 < first + remaining) {
    const uint32_t target = uint32_t(needed - 1 - first);
    uint32_t p2 = my_start & ST_OFF_MASK;
    for (uint32_t t = 0; t < target; ++t)
      p2 += lj_step<false, true, 0>(L, dp, j, p2, 0u, true) >> 10;
    a.results[s].last_slot = lb * LJ_OWN + uint32_t(j - 1);
    a.results[s].last_pos = p2;
  }
}

// ---------------------------------------------------------------------------
// K4b: end-of-stream tail.  The reference keeps decoding when the symbols run
// past the end of data: after the FFxx marker the bit reader supplies zeros
// (BitStreamerJPEG.h:155-179) until its position budget is exhausted, then
// throws "Buffer overflow read in BitStreamer" (BitStreamer.h:120-131).  With
// c* = the symbol start whose refill first touches the marker, symbols may
// start at un-stuffed bit offsets <= c* + 160 (64 cache bits + 4 more refills);
// without a marker, at offsets <= 32*floor((D+16)/4), D = data bytes.
// Only damaged / truncated streams get here: one lane re-reads the last few
// subsequences sequentially and delivers the symbols K4 could not.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lj_count_drops(const uint8_t* in, uint64_t from,
                                                   uint64_t to, int lane);
__device__ __forceinline__ uint64_t lj_drops_before(const LjArgs& a,
                                                    const LjStreamDev& S,
                                                    const uint8_t* in, uint64_t x,
                                                    int lane);

struct Sym {
  uint32_t total; // bits consumed
  uint32_t ssss;
  uint32_t code_len;
  bool ok;
};

__device__ __forceinline__ Sym lj_symbol_global(uint32_t w, const TabLds* tb) {
  const uint32_t e = lj_lut16(*tb, w >> (32 - LUT_BITS));
  if (e & 31u)
    return {e >> 10, (e >> 5) & 31u, e & 31u, true};
  for (uint32_t l = LUT_BITS + 1; l <= tb->max_len; ++l) {
    const uint32_t c = w >> (32 - l);
    const uint32_t mc = tb->max_code[l];
    if (mc != NO_CODE && c <= mc) {
      const uint32_t val = tb->values[(c - tb->val_offset[l]) & 0xFFFFu];
      const uint32_t ssss = (tb->las && val != 16u) ? (val & 15u) : val;
      const uint32_t extra = ssss == 16u ? (tb->fix16 ? 16u : 0u)
                                         : (tb->las ? ssss - (val >> 4) : ssss);
      return {l + extra, ssss, l, true};
    }
  }
  return {0, 0, 0, false};
}

__global__ __launch_bounds__(64) void lj_tail_kernel(LjArgs a) {
  const uint32_t s = blockIdx.x;
  const LjStreamDev& S = a.streams[s];
  LjResult& R = a.results[s];
  const int lane = threadIdx.x;
  const uint32_t avail = R.avail_lo;
  if (R.status != 0 || uint64_t(avail) >= S.needed)
    return; // the common case: every symbol starts inside the data
  if (!lj_legacy_takes(a, s, S))
    return; // a fused-path stream whose difference scratch does not exist yet
  const uint8_t* in = a.in_base + S.in_offset;
  const bool has_marker = R.marker_pos != 0xFFFFFFFFu && R.marker_pos < S.in_bytes;
  const uint64_t M = has_marker ? R.marker_pos : lj_data_end(S);
  const uint64_t D = M - lj_drops_before(a, S, in, M, lane);
  // start two subsequences before the one holding the last data byte
  const uint64_t ps = M > 0 ? (M - 1) / LJ_P : 0;
  const uint64_t s0 = ps >= 2 ? ps - 2 : 0;
  const uint64_t L0 = s0 * LJ_P - lj_drops_before(a, S, in, s0 * LJ_P, lane);
  if (lane != 0)
    return;
  const uint32_t st0 =
      s0 == 0 ? uint32_t(S.start_bit) : (a.sub_state[S.first_subseq + s0 - 1] & ST_MASK);
  if (st0 & ST_ERR) {
    atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));
    return;
  }
  const uint32_t b0 = uint32_t(s0 / LJ_OWN);
  uint64_t idx = a.block_base[S.first_block + b0];
  for (uint64_t q = uint64_t(b0) * LJ_OWN; q < s0; ++q)
    idx += a.sub_state[S.first_subseq + q] >> 16;

  // sequential un-stuffing reader over physical bytes [x, M), zeros afterwards
  uint64_t x = s0 * LJ_P;
  const bool raw = S.raw != 0;
  if (!raw && x < M && x > 0 && in[x] == 0x00 && in[x - 1] == 0xFF)
    ++x; // the slot starts on a stuffing byte
  uint64_t buf = 0;
  uint32_t nb = 0;
  auto refill = [&]() {
    while (nb <= 56) {
      uint32_t byte = 0;
      if (x < M) {
        // MSB32: stream byte x is byte x ^ 3 of memory (zero past the real end)
        const uint64_t mx = S.pair ? (x ^ 3) : x;
        byte = mx < S.in_bytes ? in[mx] : 0u;
        ++x;
        if (byte == 0xFF && !raw)
          ++x; // its stuffing byte (every FF before M is followed by 00)
      }
      buf |= uint64_t(byte) << (56 - nb);
      nb += 8;
    }
  };
  refill();
  uint64_t c = 8 * L0; // un-stuffed bit offset of the reader
  {
    uint32_t skip = st0 & ST_OFF_MASK;
    buf <<= skip;
    nb -= skip;
    c += skip;
  }
  uint32_t phase = (st0 >> ST_PHASE_SHIFT) & 7u;
  const bool multi = S.n_tables > 1;
  const TabLds* tabs = a.tables + S.table_base;
  const int64_t T0 = int64_t(32 * (D / 4)) - 31;
  // refill k reads bytes [4k, 4k+4) and throws once 4k > size + 2*MaxProcessBytes
  // (BitStreamer.h:125-127): 16 for BitStreamerJPEG, 8 for BitStreamerMSB
  const uint64_t limit_nomarker =
      (raw && S.raw_limit) ? S.raw_limit - 1 : 32 * ((D + (raw ? 8 : 16)) / 4);
  int64_t cstar = -1;
  int16_t* __restrict__ dst = a.diffs + S.diff_offset;
  if (S.pair) {
    // HasselbladDecompressor.cpp:87-92: decodeCodeValue x 2 (each fills 32 bits),
    // getBits(len) x 2 (each fills len bits).  Refill k reads word k and throws once
    // 4 k > size + 8 (BitStreamer.h:125-127), so a fill point at bit c that needs n
    // bits is fine iff c + n <= 32 * (floor((size + 8) / 4) + 1).
    const uint64_t lim = 32 * ((S.in_bytes + 8) / 4 + 1);
    const TabLds* tb = tabs;
    uint32_t* dst32 = reinterpret_cast<uint32_t*>(a.diffs + S.diff_offset);
    while (idx < S.needed) {
      const uint64_t c0 = c;
      uint32_t len[2], d[2];
      for (int k = 0; k < 2; ++k) {
        if (c + 32 > lim) {
          atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_INPUT_OVERFLOW));
          return;
        }
        refill();
        const Sym sy = lj_symbol_global(uint32_t(buf >> 32), tb);
        if (!sy.ok) {
          atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));
          return;
        }
        len[k] = sy.ssss;
        buf <<= sy.code_len;
        nb -= sy.code_len;
        c += sy.code_len;
      }
      for (int k = 0; k < 2; ++k) {
        d[k] = 0;
        if (len[k]) {
          if (c + len[k] > lim) {
            atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_INPUT_OVERFLOW));
            return;
          }
          refill();
          const uint32_t v = uint32_t(buf >> (64 - len[k]));
          uint32_t x2 = (v >> (len[k] - 1)) ? v : v + 1u - (1u << len[k]);
          if (len[k] == 16 && v == 0xFFFFu)
            x2 = 0x8000u; // "if (diff == 65535) return -32768" :66-67
          d[k] = x2 & 0xFFFFu;
          buf <<= len[k];
          nb -= len[k];
          c += len[k];
        }
      }
      if (idx >= avail)
        dst32[idx] = d[0] | (d[1] << 16);
      if (idx + 1 == S.needed) {
        R.tail_used = 1;
        R.last_c_lo = uint32_t(c0);
        R.last_c_hi = uint32_t(c0 >> 32);
      }
      ++idx;
    }
    R.avail_lo = uint32_t(S.needed);
    return;
  }
  while (idx < S.needed) {
    if (has_marker) {
      if (cstar < 0 && int64_t(c) >= T0)
        cstar = int64_t(c);
      if (cstar >= 0 && int64_t(c) > cstar + 160) {
        atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_INPUT_OVERFLOW));
        return;
      }
    } else if (c > limit_nomarker) {
      atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_INPUT_OVERFLOW));
      return;
    }
    refill();
    const uint32_t w = uint32_t(buf >> 32);
    const Sym sy = lj_symbol_global(w, tabs + (multi ? S.tab_of_phase[phase] : 0));
    if (!sy.ok) {
      atomicCAS(&R.status, 0u, uint32_t(RSX_ERR_BAD_HUFFMAN_CODE));
      return;
    }
    int diff;
    if (sy.ssss == 0u) {
      diff = 0;
    } else if (sy.ssss == 16u) {
      diff = -32768;
    } else {
      // (NikonLASDecompressor: only len - shl of the len bits are in the stream)
      const uint32_t nbits = sy.total - sy.code_len, shl = sy.ssss - nbits;
      const uint32_t v = nbits ? (w << sy.code_len) >> (32 - nbits) : 0u;
      diff = int((((v << 1) + 1u) << shl) >> 1);
      if ((diff & (1 << (sy.ssss - 1))) == 0)
        diff -= (1 << sy.ssss) - (shl ? 0 : 1);
    }
    if (idx >= avail)
      dst[idx] = int16_t(diff);
    if (idx + 1 == S.needed) {
      R.tail_used = 1;
      R.last_c_lo = uint32_t(c);
      R.last_c_hi = uint32_t(c >> 32);
    }
    buf <<= sy.total;
    nb -= sy.total;
    c += sy.total;
    if (multi)
      phase = (phase + 1 == S.period) ? 0u : phase + 1;
    ++idx;
  }
  R.avail_lo = uint32_t(S.needed);
}

// ---------------------------------------------------------------------------
// K7: decode()/decompress() return value = BitStreamerJPEG::getStreamPosition()
// after the last decoded symbol: let c be the un-stuffed bit
// offset at which the last decoded symbol starts; K = ceil(c/32)+1 refills of 4
// data bytes have happened; D = data bytes before the end marker.  If 4K > D
// the answer is the marker's offset, otherwise the physical offset just past
// the first 4K data bytes.  One wavefront per stream.
// ---------------------------------------------------------------------------
// stuffing bytes (00 preceded by FF) at stream positions [from, to), one wave;
// each lane scans 16-byte pieces (byte loads of a 17-byte window hit L1/L2)

__device__ __forceinline__ uint32_t lj_count_drops(const uint8_t* in, uint64_t from,
                                                   uint64_t to, int lane) {
  uint32_t n = 0;
  // (four 16-byte pieces of a lane in flight at once: the loop is as long as its loads'
  // latencies in a row -- 16 of them for a workgroup's 16 KB, 10 us of lj_scan_kernel's 40)
  for (uint64_t q0 = from + uint64_t(lane) * 16; q0 < to; q0 += 4 * 64 * 16) {
    uint4 v[4];
    uint32_t pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t p0 = q0 + uint64_t(u) * 64 * 16;
      v[u] = make_uint4(0, 0, 0, 0);
      pv[u] = 0;
      if (p0 + 16 <= to) {
        // one 16-byte load (global loads take unaligned addresses)
        __builtin_memcpy(&v[u], in + p0, 16);
        pv[u] = p0 > 0 ? in[p0 - 1] : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t p0 = q0 + uint64_t(u) * 64 * 16;
      if (p0 >= to)
        continue;
      uint32_t prev = pv[u];
      if (p0 + 16 <= to) {
        // a stuffing byte is a zero byte whose predecessor is FF
        const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t z = lj_zero_bytes(d[k]), f = lj_zero_bytes(~d[k]);
          n += uint32_t(__builtin_popcount(z & ((f << 8) | (prev == 0xFFu ? 0x80u : 0u))));
          prev = d[k] >> 24;
        }
        continue;
      }
      prev = p0 > 0 ? in[p0 - 1] : 0u;
      for (uint64_t p = p0; p < to; ++p) {
        const uint32_t c = in[p];
        n += (c == 0u && prev == 0xFFu) ? 1u : 0u;
        prev = c;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    n += __shfl_down(n, o, 64);
  return __shfl(n, 0, 64);
}

// stuffing bytes before stream position x (x <= end of data)
__device__ __forceinline__ uint64_t lj_drops_before(const LjArgs& a,
                                                    const LjStreamDev& S,
                                                    const uint8_t* in, uint64_t x,
                                                    int lane) {
  if (S.raw)
    return 0;
  uint64_t lb = x / LJ_R;
  if (lb >= S.n_blocks)
    lb = S.n_blocks - 1;
  return uint64_t(a.block_drop_base[S.first_block + lb]) +
         lj_count_drops(in, lb * LJ_R, x, lane);
}

__device__ void lj_consumed_body(const LjArgs& a, uint32_t s, int lane) {
  const LjStreamDev& S = a.streams[s];
  LjResult& R = a.results[s];
  if (!lj_bookkeeping_takes(a, s, S))
    return;
  if (R.status != 0 || uint64_t(R.avail_lo) < S.needed)
    return;
  const uint8_t* in = a.in_base + S.in_offset;
  const bool has_marker = R.marker_pos != 0xFFFFFFFFu && R.marker_pos < S.in_bytes;
  const uint64_t M = has_marker ? R.marker_pos : lj_data_end(S);
  // un-stuffed bit offset of the last symbol's start
  const uint64_t slot_phys = uint64_t(R.last_slot) * LJ_P;
  const uint64_t drops_slot = lj_drops_before(a, S, in, slot_phys, lane);
  uint64_t c = (slot_phys - drops_slot) * 8 + R.last_pos;
  if (R.tail_used)
    c = (uint64_t(R.last_c_hi) << 32) | R.last_c_lo;
  if (S.raw) {
    // where the next symbol would start: the last one's start + its length
    // (bytes past the end of the buffer read as zero)
    if (lane == 0) {
      // the 32 stream bits at bit offset cc (zero past the end of the buffer)
      auto window = [&](uint64_t cc) -> uint32_t {
        const uint64_t by = cc >> 3;
        uint64_t acc = 0;
        for (int i = 0; i < 5; ++i) {
          const uint64_t q = by + i, mq = S.pair ? (q ^ 3) : q;
          acc = (acc << 8) | (mq < S.in_bytes ? in[mq] : 0u);
        }
        return uint32_t((acc << (cc & 7)) >> 8);
      };
      const TabLds* tb = a.tables + S.table_base;
      const Sym sy = lj_symbol_global(window(c), tb);
      uint64_t end = c + (sy.ok ? sy.total : 0u);
      if (S.pair && sy.ok) {
        // [code1][code2][bits1][bits2]: sy.total covers code1 + bits1
        const Sym s2 = lj_symbol_global(window(c + sy.code_len), tb);
        end += s2.ok ? s2.total : 0u;
        // BitStreamer::getStreamPosition: pos - (fillLevel >> 3) = ceil(end / 8)
        R.consumed = uint32_t((end + 7) / 8);
      }
      R.end_lo = uint32_t(end);
      R.end_hi = uint32_t(end >> 32);
    }
    return;
  }
  const uint64_t K = (c + 31) / 32 + 1;
  // (the last symbol's slot and the end of data lie in the same workgroup's region as a
  // rule: the stuffing bytes between them are all that is left to count)
  uint64_t drops_M;
  {
    uint64_t lbs = slot_phys / LJ_R, lbm = M / LJ_R;
    if (lbs >= S.n_blocks)
      lbs = S.n_blocks - 1;
    if (lbm >= S.n_blocks)
      lbm = S.n_blocks - 1;
    drops_M = (!S.raw && lbs == lbm && slot_phys <= M)
                  ? drops_slot + lj_count_drops(in, slot_phys, M, lane)
                  : lj_drops_before(a, S, in, M, lane);
  }
  const uint64_t D = M - drops_M;
  uint64_t result;
  if (4 * K > D) {
    // the last refill touched the marker (or ran off the end of the buffer)
    result = has_marker ? M : S.in_bytes + (4 * K - D);
  } else {
    // physical offset just past the first 4K data bytes
    const uint64_t target = 4 * K;
    // workgroup region whose logical start lb*R - drop_base[lb] is <= target
    uint32_t lo = 0, hi = S.n_blocks - 1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      const uint64_t ls = uint64_t(mid) * LJ_R - a.block_drop_base[S.first_block + mid];
      if (ls <= target)
        lo = mid;
      else
        hi = mid - 1;
    }
    const uint64_t x0 = uint64_t(lo) * LJ_R;
    const uint64_t logical0 = x0 - a.block_drop_base[S.first_block + lo];
    // Inside the region (<= 16 KiB; only bottom-overhanging tiles get here): every
    // lane counts the data bytes of its 256-byte piece, a wave scan finds the piece
    // the target falls into, and that lane alone walks it.  (One lane walking the
    // whole region took 1.7 ms on an 8189x5462 DNG.)
    const uint64_t xs = x0 + 256u * uint32_t(lane);
    uint64_t xe = xs + 256u;
    if (xe > M)
      xe = M;
    // (16 bytes at a time, four loads in flight: byte by byte the 256 bytes of a lane were
    // 512 loads in a row -- 50 us of an overhanging tile's 270)
    auto drops16 = [&](uint64_t p0, uint4 v, uint32_t prev) -> uint32_t {
      (void)p0;
      const uint32_t d[4] = {v.x, v.y, v.z, v.w};
      uint32_t n = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t z = lj_zero_bytes(d[k]), f = lj_zero_bytes(~d[k]);
        n += uint32_t(__builtin_popcount(z & ((f << 8) | (prev == 0xFFu ? 0x80u : 0u))));
        prev = d[k] >> 24;
      }
      return n;
    };
    uint32_t data = 0;
    for (uint64_t q0 = xs; q0 < xe; q0 += 64) {
      uint4 v[4];
      uint32_t pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t p0 = q0 + 16u * uint32_t(u);
        v[u] = make_uint4(0, 0, 0, 0);
        pv[u] = 0;
        if (p0 + 16 <= xe) {
          __builtin_memcpy(&v[u], in + p0, 16);
          pv[u] = p0 > 0 ? in[p0 - 1] : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t p0 = q0 + 16u * uint32_t(u);
        if (p0 >= xe)
          continue;
        if (p0 + 16 <= xe) {
          data += 16u - drops16(p0, v[u], pv[u]);
          continue;
        }
        for (uint64_t q = p0; q < xe; ++q)
          data += (in[q] == 0x00 && q > 0 && in[q - 1] == 0xFF) ? 0u : 1u;
      }
    }
    const uint32_t incl = lj_wave_scan(data, lane);
    const uint64_t want = target - logical0; // data bytes of the region still to pass (>= 0)
    // the first piece whose running count reaches `want` (the last one if none does)
    const bool mine_or_later = uint64_t(incl) >= want;
    const uint64_t ballot = __ballot(mine_or_later);
    const int owner = ballot ? __builtin_ctzll(ballot) : 63;
    uint64_t x = xs;
    uint64_t logical = logical0 + incl - data;
    if (lane == owner) {
      // whole 16-byte pieces first, then the bytes of the piece the target falls into
      while (x + 16 <= M && x + 16 <= xs + 256u) {
        uint4 v;
        __builtin_memcpy(&v, in + x, 16);
        const uint32_t d16 = 16u - drops16(x, v, x > 0 ? in[x - 1] : 0u);
        if (logical + d16 >= target)
          break;
        logical += d16;
        x += 16;
      }
      while (logical < target && x < M) {
        const bool drop = in[x] == 0x00 && x > 0 && in[x - 1] == 0xFF;
        if (!drop)
          ++logical;
        ++x;
      }
    }
    x = uint64_t(__shfl(uint32_t(x), owner, 64)) |
        (uint64_t(__shfl(uint32_t(x >> 32), owner, 64)) << 32);
    // a data FF is consumed together with its stuffing byte
    // (BitStreamerJPEG.h:145-151)
    if (x < M && x > 0 && in[x - 1] == 0xFF && in[x] == 0x00)
      ++x;
    result = x;
  }
  if (lane == 0)
    R.consumed = uint32_t(result);
}

__global__ __launch_bounds__(64) void lj_consumed_kernel(LjArgs a) {
  lj_consumed_body(a, blockIdx.x, int(threadIdx.x));
}

// ---------------------------------------------------------------------------
// Restart intervals (LJpegDecompressor.cpp:276-298): every FF xx (xx != 00) in
// the scan is a marker; interval i+1 starts 2 bytes after the i-th one.  This
// kernel lists them (unordered; the host sorts the handful of entries).
// ---------------------------------------------------------------------------
// the results of a run, before it: marker_pos = 0xFFFFFFFF (an atomicMin target), everything else 0
__global__ __launch_bounds__(256) void lj_init_results_kernel(LjResult* results, uint32_t n) {
  static_assert(sizeof(LjResult) % 4 == 0, "dwords");
  constexpr uint32_t DW = uint32_t(sizeof(LjResult) / 4);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n * DW)
    return;
  const uint32_t k = i / DW, f = i - k * DW;
  reinterpret_cast<uint32_t*>(results + k)[f] =
      f == uint32_t(offsetof(LjResult, marker_pos) / 4) ? 0xFFFFFFFFu : 0u;
}

__global__ __launch_bounds__(256) void lj_marker_scan_kernel(
    const uint8_t* __restrict__ in, uint64_t bytes, uint32_t* count, uint2* list,
    uint32_t cap) {
  const uint64_t off = (uint64_t(blockIdx.x) * 256 + threadIdx.x) * 16;
  if (off >= bytes)
    return;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const uint64_t p = off + b;
    if (p + 1 < bytes && in[p] == 0xFF && in[p + 1] != 0x00) {
      const uint32_t i = atomicAdd(count, 1u);
      if (i < cap)
        list[i] = make_uint2(uint32_t(p), uint32_t(in[p + 1]));
    }
  }
}

// ---------------------------------------------------------------------------
// Restart intervals without a host round trip (round 5; LJpegDecompressor.cpp:277-335).
// Every interval is a stream of a child plan; where an interval starts and how long it is
// is only known once the RSTn markers have been found.  Until round 5 the marker list went
// to the host, which built (or re-validated) the child plan and launched it: one
// synchronisation in the middle of every decode, 0.2 of the 0.41 ms of a cfg-4 frame.  Now
// the child plan is made ONCE, from the geometry alone, with room for the blocks ANY marker
// placement needs (sum of ceil(bytes_i / LJ_R) <= bytes / LJ_R + intervals + 2), and two
// small kernels write its stream records, its block -> stream map and its ticket order
// from the marker list on the device; K0 and the single-pass kernel follow in the same
// stream.  The host looks at markers, statuses and RSTn numbering when it fetches the
// results -- and redoes the job the old way (exact child plan, host-built) whenever
// something is off: markers missing, more FFxx than the list holds, a stream the
// single-pass kernel gave up on.
// ---------------------------------------------------------------------------
struct DriJobDev {
  uint64_t in_offset, in_bytes; // the job's scan data
  uint32_t n_ri;                // restart intervals = streams of the child plan
  uint32_t first_stream;        // its first child stream
  uint32_t list_off, cap;       // its slice of the (unsorted) marker list
  uint32_t sorted_off;          // its slice of the sorted list (n_ri entries)
  uint32_t pad_;
};
constexpr uint32_t DRI_NONE = 0xFFFFFFFFu;

// The marker scan of ALL jobs in one launch, sixteen bytes a lane as ONE load on the
// buffer's 16-byte grid (lj_marker_scan_kernel reads byte by byte, one launch per job: 70 of
// the 89 us the device path's first version spent in front of K0 on four 11 MB tiles).  Only
// lanes whose bytes hold an FF (one in sixteen) look at them one by one.
__global__ __launch_bounds__(256) void lj_dri_scan_kernel(const uint8_t* __restrict__ in_base,
                                                          const DriJobDev* jobs, uint32_t* count,
                                                          uint2* list) {
  const DriJobDev J = jobs[blockIdx.y];
  const uint8_t* in = in_base + J.in_offset;
  const uint64_t bytes = J.in_bytes;
  const uint32_t lead = uint32_t(reinterpret_cast<uintptr_t>(in) & 15u);
  // chunk c = stream bytes [16 c - lead, 16 c - lead + 16)
  const int64_t p0 = int64_t(uint64_t(blockIdx.x) * 256 + threadIdx.x) * 16 - int64_t(lead);
  if (p0 >= int64_t(bytes))
    return;
  uint32_t w[5] = {0, 0, 0, 0, 0}; // the chunk and the byte behind it
  if (p0 >= 0 && p0 + 17 <= int64_t(bytes)) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + p0);
    w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    if (!(has_ff(w[0]) | has_ff(w[1]) | has_ff(w[2]) | has_ff(w[3])))
      return;
    w[4] = in[p0 + 16];
  } else {
    for (int i = 0; i < 17; ++i) {
      const int64_t p = p0 + i;
      const uint32_t b = (p >= 0 && p < int64_t(bytes)) ? in[p] : 0u;
      w[i >> 2] |= b << (8 * (i & 3));
    }
  }
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const uint32_t x = (w[b >> 2] >> (8 * (b & 3))) & 0xFFu;
    const uint32_t y = (w[(b + 1) >> 2] >> (8 * ((b + 1) & 3))) & 0xFFu;
    const int64_t p = p0 + b;
    if (x == 0xFFu && y != 0u && p >= 0 && p + 1 < int64_t(bytes)) {
      const uint32_t i = atomicAdd(&count[blockIdx.y], 1u);
      if (i < J.cap)
        list[J.list_off + i] = make_uint2(uint32_t(p), y);
    }
  }
}

// every CU drops its scalar data cache (rsx_ljpeg_dev.h, lj_fresh_scalars' note)
__global__ __launch_bounds__(64) void lj_dcache_inv_kernel() {
  asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// the first n_ri markers of every job in stream order: rank of an entry = entries in front
__global__ __launch_bounds__(256) void lj_dri_sort_kernel(const DriJobDev* jobs,
                                                          const uint32_t* count,
                                                          const uint2* list, uint2* sorted) {
  const DriJobDev J = jobs[blockIdx.y];
  const uint32_t n = min(count[blockIdx.y], J.cap);
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= n)
    return;
  const uint2* L = list + J.list_off;
  const uint2 me = L[e];
  uint32_t rank = 0;
  for (uint32_t k = 0; k < n; ++k)
    rank += L[k].x < me.x ? 1u : 0u; // (positions are distinct)
  if (rank < J.n_ri)
    sorted[J.sorted_off + rank] = me;
}

// stream records, block -> stream map and ticket order of the child plan.  One workgroup.
// status[d]: 1 markers missing, 2 more FFxx than the list holds; status[n_jobs]: 4 the
// blocks do not fit the plan (cannot happen with its bound; checked all the same).
__global__ __launch_bounds__(1024) void lj_dri_layout_kernel(
    LjStreamDev* streams, uint32_t n_streams, uint32_t* block_stream, uint4* fast_order,
    uint32_t grid_blocks, const DriJobDev* jobs, uint32_t n_jobs, const uint32_t* count,
    const uint2* sorted, uint32_t* status) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t total_s;
  const uint32_t tid = threadIdx.x;
  // (cross-lane traffic through global memory inside ONE workgroup: agent-scope accesses
  // bypass the CU's vector cache)
  auto ld = [](const uint32_t* q) {
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto st = [](uint32_t* q, uint32_t v) {
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (uint32_t d = 0; d < n_jobs; ++d) {
    const DriJobDev J = jobs[d];
    const uint32_t c = count[d], n = min(c, J.cap);
    const bool bad = c > J.cap || n + 1u < J.n_ri;
    if (tid == 0 && bad)
      status[d] = c > J.cap ? 2u : 1u;
    for (uint32_t i = tid; i < J.n_ri; i += 1024u) {
      LjStreamDev& S = streams[J.first_stream + i];
      uint64_t start = 0, end = 0;
      if (!bad) {
        start = i ? uint64_t(sorted[J.sorted_off + i - 1].x) + 2u : 0u;
        // up to and including the closing marker (the reference hands every interval the
        // whole remaining buffer; >= 8 bytes: BitStreamer.h:58-59)
        end = (i + 1 < J.n_ri) ? uint64_t(sorted[J.sorted_off + i].x) + 2u : J.in_bytes;
        if (end < start + 8u)
          end = start + 8u;
        if (end > J.in_bytes)
          end = J.in_bytes;
        if (start > end)
          start = end;
      }
      S.in_offset = J.in_offset + start;
      S.in_bytes = end - start;
      st(&S.n_blocks, uint32_t((end - start + uint64_t(LJ_R) - 1u) / uint64_t(LJ_R)));
    }
  }
  __syncthreads();
  // exclusive scan of the block counts: a contiguous chunk of streams per lane
  const uint32_t chunk = (n_streams + 1023u) / 1024u;
  const uint32_t lo = min(tid * chunk, n_streams), hi = min(lo + chunk, n_streams);
  uint32_t sum = 0;
  for (uint32_t k = lo; k < hi; ++k)
    sum += ld(&streams[k].n_blocks);
  part[tid] = sum;
  __syncthreads();
  for (uint32_t o = 1; o < 1024u; o <<= 1) {
    const uint32_t v = tid >= o ? part[tid - o] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  if (tid == 1023u)
    total_s = part[1023];
  __syncthreads();
  const bool fits = total_s <= grid_blocks;
  if (tid == 0 && !fits)
    status[n_jobs] = 4u;
  uint32_t base = part[tid] - sum;
  for (uint32_t k = lo; k < hi; ++k) {
    LjStreamDev& S = streams[k];
    const uint32_t nb = fits ? ld(&S.n_blocks) : 0u;
    if (!fits) {
      st(&S.n_blocks, 0u);
      S.in_bytes = 0;
    }
    st(&S.first_block, base);
    S.first_subseq = base * uint32_t(LJ_OWN);
    base += nb;
  }
  for (uint32_t b = tid; b < grid_blocks; b += 1024u) {
    block_stream[b] = DRI_NONE;
    fast_order[b] = make_uint4(b, DRI_NONE, 0u, 0u);
  }
  __syncthreads();
  // every stream writes its own blocks: ticket = block (a stream's workgroups wait for the
  // few in front of them: no need to interleave streams of a handful of blocks each)
  for (uint32_t k = tid; k < n_streams; k += 1024u) {
    const LjStreamDev& S = streams[k];
    const uint32_t fb = ld(&S.first_block), nb = ld(&S.n_blocks);
    const uint32_t tz = lf_table_word(S.table_base, S.tab_of_phase[0], S.tab_of_phase[1],
                                      S.tab_of_phase[2], S.tab_of_phase[3]);
    for (uint32_t q = 0; q < nb; ++q) {
      block_stream[fb + q] = k;
      fast_order[fb + q] = make_uint4(fb + q, k, tz, fb);
    }
  }
}

} // namespace
// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
constexpr int RSX_INTERNAL_RETRY = -1000; // a device-laid-out child plan: redo the old way
struct LJpegPlan {
  rsx_ctx* ctx = nullptr;
  int n_jobs = 0;
  std::vector<int32_t> job_status;      // validation results
  std::vector<int> job_first_stream;    // -1 if skipped
  std::vector<int> job_n_streams;
  std::vector<LjStreamDev> streams;
  uint32_t total_blocks = 0, total_subseq = 0, total_rows = 0;
  uint64_t total_diffs = 0;
  int max_tables = 1;
  // kernel classes of the legacy route (difference scratch + K5 / K6): `legacy` for
  // the streams that always take it, `fallback` for the fused-path streams should
  // they turn out to be damaged
  struct Classes {
    bool plain = false, las = false, pair = false, multi = false;
    bool comp[7] = {}; // [1..4] interleaved n_comp; [5], [6]: sRaw groups of 4, 6
    bool nikon = false, sony = false;
  };
  Classes legacy, fallback;
  // which instantiations the plan's streams need
  bool sync_present[2][5] = {};   // [several tables][fused-path components, 0 = legacy]
  bool direct_present[2][5] = {}; // fused path: [several tables][components]
  bool any_direct = false, any_legacy = false;
  bool any_fast_legacy = false; // legacy-route streams on the single-pass kernel (3 components)
  bool legacy_fallback_ready = false; // difference scratch of the fused streams allocated
  // single-pass path (rsx_ljpeg_fast.hip)
  bool fast_present[3][5] = {}; // [one table / two alternating / a table per phase][components]
  bool any_fast_mt = false;     // some stream takes its two-table instantiation
  bool any_fast_pt = false;     // ... its table-per-phase instantiation (LjStreamDev::fast == 3)
  uint32_t pt_np = 2;           // the most phases such a stream has (sizes K0's LDS)
  bool any_fast_nk = false;     // ... writes the pixels of a Nikon-type stream (fast_nk)
  bool any_fast_diffs = false;  // ... leaves differences for the legacy reconstruction (fast_diffs)
  DeviceBuffer d_k0e;           // K0's hand-over words (LjArgs::k0e)
  DeviceBuffer d_k0p;           // K0's phase look-back words (LjArgs::k0p), table-per-phase plans
  std::vector<Cr2Strip> h_strips; // (host copy: which pixels a prefix of a stream's symbols completes)
  LjArgs run_args{};            // the run in progress (ljpeg_plan_run_begin .. _end)
  bool any_fast = false;       // some stream takes the single-pass kernel
  uint32_t fast_lds = 0;       // LDS bytes of its launches
  uint32_t fast_uniform_nb = 0, fast_rotate = 0; // (LjArgs)
  uint64_t results_clean_for = 0;          // the run (run_count) whose results K0 has cleared ...
  hipStream_t results_clean_stream = nullptr; // ... on this stream
  std::vector<uint8_t> slow_strikes; // per stream: consecutive runs it went to the slow path
  // streams taken off the single-pass kernel after two such runs: the value `fast` had and
  // the run in which it was cleared (0: not demoted).  The demotion DECAYS: a cached plan
  // sees other images later (host-pointer calls), and one pair of hard frames must not
  // send every later frame of that layout to the slower pipeline for good.
  std::vector<uint8_t> demoted_fast;
  std::vector<uint32_t> demoted_at;
  static constexpr uint32_t DEMOTION_RUNS = 8;
  bool any_pipeline = false;   // some stream takes the multi-kernel pipeline in the first pass
  bool expect_slow = false;    // the last run left FL_SLOW streams: launch the second pass at once
  bool slow_pass_launched = false; // ... this run already has
  DeviceBuffer d_fast_tabs, d_fast_tabs16, d_lb, d_tickets, d_fast_order, d_fast_z, d_fast_level, d_k0w,
      d_block_base0;
  uint32_t run_count = 0;      // runs so far (parity: which level word a run uses)
  uint32_t level_mask = 7;     // LDS levels the single-pass kernel is launched at (LjArgs::fast_level_mask)
  uint32_t h_level[4] = {};
  DeviceBuffer d_dbg; // experiment builds: phase time stamps of the single-pass kernel
  DeviceBuffer d_block_flags, d_block_tf, d_sub_start;
  DeviceBuffer d_streams, d_tables, d_block_stream, d_strips, d_sub_state, d_sub_sums,
      d_sub_first, d_sub_psum,
      d_block_start, d_block_exit, d_block_sum, d_block_base, d_block_psum, d_block_pbase,
      d_block_drops, d_block_drop_base, d_results, d_diffs, d_vseed, d_row_edge, d_unstuffed;
  KernelTimer* timer = nullptr; // set for the duration of a timed run
  std::vector<LjResult> h_results;
  int stitch_rounds = 2;
  const void* last_in = nullptr;
  void* last_out = nullptr;
  int extra_stitch_rounds = 0; // statistics: rounds needed beyond the default
  // jobs with restart intervals: their streams are only known once the RSTn
  // markers have been located on the device (one host round trip per run)
  struct DriJob {
    int job = 0;
    LJpegJobIn in;
    std::vector<rsx_huff_table> tables;
    uint32_t rows_per_ri = 0, n_ri = 0;
    std::vector<uint32_t> starts;  // interval start offsets (job-relative)
    std::vector<uint32_t> markers; // sorted marker offsets (job-relative)
    std::vector<uint8_t> codes;
    int status = RSX_OK;
  };
  std::vector<DriJob> dri;
  DeviceBuffer d_marker_count, d_marker_list;
  std::vector<uint32_t> dri_signature; // marker layout the child plan was built for
  LJpegPlan* child = nullptr;          // one stream per restart interval
  // ... and its device-laid-out sibling (round 5): made once from the geometry, its stream
  // records written by lj_dri_layout_kernel in every run
  LJpegPlan* child_dev = nullptr;
  std::vector<std::pair<int, int>> child_dev_owner;
  bool child_dev_failed = false; // this plan's jobs do not fit the scheme: host path only
  bool dri_ran_dev = false;      // the last run took the device path
  bool dev_layout = false;       // (on the child) its layout lives on the device
  std::vector<DriJobDev> dri_dev;
  DeviceBuffer d_dri_jobs, d_dri_sorted, d_dri_status;
  std::vector<std::pair<int, int>> child_owner; // child job -> (dri index, interval)
  // NikonDecompressor streams
  bool any_nikon = false, any_pair = false, any_multi = false, any_lut11 = false;
  std::vector<NkStreamDev> nk;         // parallel to streams
  DeviceBuffer d_nk, d_nk_tables, d_nk_rowpow, d_nk_pup;
  DeviceBuffer d_transfer; // fallback path only (allocated on first use)
  // jobs with a split row: the rows after it are a second stream (other table)
  // that starts at the bit where the first part ends -- known after it ran
  struct NkSplit {
    int job = 0;
    int stream = 0; // first part
    LJpegJobIn in;
    rsx_huff_table table{};
    int status = RSX_OK;
  };
  std::vector<NkSplit> nk_split;
  std::vector<uint64_t> nk_signature; // end bits the split child was built for
  LJpegPlan* nk_child = nullptr;
  std::vector<int> nk_child_owner;    // child job -> nk_split index
};

namespace {

// the results of the current run (two sets taking turns: lj_unstuff_kernel clears the next run's)
LjResult* results_of_run(const LJpegPlan* p) {
  return static_cast<LjResult*>(p->d_results.ptr) + size_t(p->run_count & 1u) * p->streams.size();
}

LjArgs make_args(LJpegPlan* p, const void* in_dev, void* out_dev) {
  LjArgs a{};
  a.blk0 = 0;
  a.blk_n = a.n_blocks_plan = p->total_blocks;
  a.in_base = static_cast<const uint8_t*>(in_dev);
  a.out_base = static_cast<uint8_t*>(out_dev);
  a.streams = static_cast<const LjStreamDev*>(p->d_streams.ptr);
  a.tables = static_cast<const TabLds*>(p->d_tables.ptr);
  a.block_stream = static_cast<const uint32_t*>(p->d_block_stream.ptr);
  a.strips = static_cast<const Cr2Strip*>(p->d_strips.ptr);
  a.sub_state = static_cast<uint32_t*>(p->d_sub_state.ptr);
  a.sub_sums = static_cast<uint2*>(p->d_sub_sums.ptr);
  a.sub_first = static_cast<uint32_t*>(p->d_sub_first.ptr);
  a.sub_psum = static_cast<uint2*>(p->d_sub_psum.ptr);
  a.block_flags = static_cast<uint32_t*>(p->d_block_flags.ptr);
  a.block_tf = static_cast<uint16_t*>(p->d_block_tf.ptr);
  a.sub_start = static_cast<uint16_t*>(p->d_sub_start.ptr);
  a.block_psum = static_cast<uint2*>(p->d_block_psum.ptr);
  a.block_pbase = static_cast<uint2*>(p->d_block_pbase.ptr);
  a.row_edge = static_cast<uint4*>(p->d_row_edge.ptr);
  a.block_start = static_cast<uint32_t*>(p->d_block_start.ptr);
  a.block_exit = static_cast<uint32_t*>(p->d_block_exit.ptr);
  a.block_sum = static_cast<uint32_t*>(p->d_block_sum.ptr);
  a.block_base = static_cast<uint32_t*>(p->d_block_base.ptr);
  a.block_drops = static_cast<uint32_t*>(p->d_block_drops.ptr);
  a.block_drop_base = static_cast<uint32_t*>(p->d_block_drop_base.ptr);
  a.unstuffed = static_cast<uint4*>(p->d_unstuffed.ptr);
  // (two sets of results, taking turns: K0 clears the next run's, see there)
  a.results = results_of_run(p);
  a.results_next = static_cast<LjResult*>(p->d_results.ptr) +
                   size_t((p->run_count & 1u) ^ 1u) * p->streams.size();
  a.diffs = static_cast<int16_t*>(p->d_diffs.ptr);
  a.vseed = static_cast<uint16_t*>(p->d_vseed.ptr);
  a.n_streams = uint32_t(p->streams.size());
  a.total_rows = p->total_rows;
  a.nk = static_cast<const NkStreamDev*>(p->d_nk.ptr);
  a.nk_tables = static_cast<const uint32_t*>(p->d_nk_tables.ptr);
  a.nk_rowpow = static_cast<const uint32_t*>(p->d_nk_rowpow.ptr);
  a.nk_pup = static_cast<int32_t*>(p->d_nk_pup.ptr);
  a.transfer = static_cast<uint16_t*>(p->d_transfer.ptr);
  a.fast_tabs = static_cast<const uint2*>(p->d_fast_tabs.ptr);
  a.fast_tabs16 = static_cast<const uint16_t*>(p->d_fast_tabs16.ptr);
  a.lb = static_cast<unsigned long long*>(p->d_lb.ptr);
  a.k0w = static_cast<unsigned long long*>(p->d_k0w.ptr);
  a.k0e = static_cast<uint32_t*>(p->d_k0e.ptr);
  a.k0p = static_cast<uint32_t*>(p->d_k0p.ptr);
  a.pt_np = p->pt_np;
#ifdef RSX_NO_K0_CHAIN // (experiments)
  a.k0_chain = 0u;
#else
  a.k0_chain = ((p->any_fast_mt || p->any_fast_pt) && p->d_k0e.ptr) ? 1u : 0u;
#endif
  a.block_base0 = static_cast<uint32_t*>(p->d_block_base0.ptr);
  a.fast_uniform_nb = p->fast_uniform_nb;
  a.dev_layout = p->dev_layout ? 1u : 0u;
  a.fast_rotate = p->fast_rotate;
  a.tickets = static_cast<uint32_t*>(p->d_tickets.ptr);
  a.fast_lds = p->fast_lds;
  {
    // LDS levels: the plan's (from the streams' average symbols per workgroup, 4 per CU
    // as a rule), then 3 and 2 workgroups per CU
    const uint32_t steps[2] = {53760u, 65536u};
    a.fast_lds_lv[0] = p->fast_lds;
    for (int l = 1; l < 3; ++l) {
      a.fast_lds_lv[l] = a.fast_lds_lv[l - 1];
      for (uint32_t c : steps)
        if (c > a.fast_lds_lv[l - 1]) {
          a.fast_lds_lv[l] = c;
          break;
        }
    }
    for (int l = 0; l < 3; ++l)
      a.fast_cap_lv[l] = a.fast_lds_lv[l] ? ljpeg_fast_stage_cap(a.fast_lds_lv[l]) : 0u;
  }
  a.run_parity = p->run_count & 1u;
  a.fast_level_mask = p->level_mask & (1u | (a.fast_lds_lv[1] != a.fast_lds_lv[0] ? 2u : 0u) |
                                       (a.fast_lds_lv[2] != a.fast_lds_lv[1] ? 4u : 0u));
  a.fast_z = static_cast<const uint32_t*>(p->d_fast_z.ptr);
  a.fast_level = static_cast<uint32_t*>(p->d_fast_level.ptr);
  // Three slots, whatever the number of streams.  (Round 3's first version took two for
  // batches of 32 streams and more -- a wrong guess delays only the workgroups of ITS stream
  // -- but every re-decode round still costs its workgroup 6 us and the ones behind it their
  // wait: 256 cfg-5 frames 348 GPix/s with three slots, 318 with two.)
  a.guess_slots = 3u;
#ifdef RSX_EXPERIMENT
  if (const char* e = getenv("RSX_GUESS_SLOTS"))
    a.guess_slots = uint32_t(atoi(e));
#endif
  a.fast_order = static_cast<const uint4*>(p->d_fast_order.ptr);
  a.dbg = static_cast<unsigned long long*>(p->d_dbg.ptr);
  a.pass = 0;
  a.fuse_consumed = 0;
  return a;
}

void mark(LJpegPlan* p, const char* name) {
  if (p->timer)
    p->timer->mark(name);
}

#ifndef RSX_K1_LDS_PAD
#define RSX_K1_LDS_PAD 0 // (experiments: unused LDS that limits K1's workgroups per CU)
#endif
template <bool STITCH, bool MULTI, int NS>
void launch_sync_one(LJpegPlan* p, const LjArgs& a, hipStream_t s) {
  if (!p->sync_present[MULTI ? 1 : 0][NS])
    return;
  using TB = SyncTable<MULTI, false>;
  const size_t lds = lj_sync_lds_bytes<TB>(MULTI ? p->max_tables : 1, LJ_BW_SYNC, NS) +
                     (STITCH ? lj_periodic_bytes() : RSX_K1_LDS_PAD);
  hipLaunchKernelGGL((lj_sync_kernel<STITCH, MULTI, false, NS>), dim3(p->total_blocks),
                     dim3(LJ_T), lds, s, a);
  mark(p, STITCH ? "lj_sync_kernel<stitch>" : "lj_sync_kernel");
}

template <bool STITCH>
void launch_sync(LJpegPlan* p, const LjArgs& a, hipStream_t s) {
  launch_sync_one<STITCH, false, 0>(p, a, s);
  launch_sync_one<STITCH, false, 1>(p, a, s);
  launch_sync_one<STITCH, false, 2>(p, a, s);
  launch_sync_one<STITCH, false, 4>(p, a, s);
  launch_sync_one<STITCH, true, 0>(p, a, s);
  launch_sync_one<STITCH, true, 1>(p, a, s);
  launch_sync_one<STITCH, true, 2>(p, a, s);
  launch_sync_one<STITCH, true, 4>(p, a, s);
  if (p->any_pair) {
    using TB = SyncTable<false, true>;
    const size_t lds =
        lj_sync_lds_bytes<TB>(1, LJ_BW_SYNC_PAIR, 0); // (no class tables for pair symbols)
    hipLaunchKernelGGL((lj_sync_kernel<STITCH, false, true, 0>), dim3(p->total_blocks),
                       dim3(LJ_T), lds, s, a);
    mark(p, STITCH ? "lj_sync_kernel<stitch,pair>" : "lj_sync_kernel<pair>");
  }
}

void launch_decode(LJpegPlan* p, const LJpegPlan::Classes& c, const LjArgs& a,
                   hipStream_t s) {
  // K4's occupancy optimum moved with its store pattern (per 4 cfg-3 frames).  One
  // 16-byte store per group: 5 workgroups per CU 217 us, 4: 199, 3: 217, 2: 258 --
  // more resident workgroups meant more half-written lines than the L2 holds.  With
  // the 64-byte bursts: 5 per CU 188 us, 4: 195.  29.5 KB = 24 granules = 5 per CU.
  constexpr size_t k4_lds = lj_lds_bytes(1, LJ_BW_DEC);
  if (c.plain) {
    hipLaunchKernelGGL((lj_decode_kernel<false>), dim3(p->total_blocks), dim3(LJ_T),
                       k4_lds, s, a);
    mark(p, "lj_decode_kernel");
  }
  if (c.las) {
    hipLaunchKernelGGL((lj_decode_kernel<false, true>), dim3(p->total_blocks), dim3(LJ_T),
                       k4_lds, s, a);
    mark(p, "lj_decode_kernel<las>");
  }
  if (c.pair) {
    hipLaunchKernelGGL(lj_decode_pair_kernel, dim3(p->total_blocks), dim3(LJ_T),
                       lj_lds_bytes(1), s, a);
    mark(p, "lj_decode_pair_kernel");
  }
  if (c.multi) {
    hipLaunchKernelGGL((lj_decode_kernel<true>), dim3(p->total_blocks), dim3(LJ_T),
                       lj_lds_bytes(p->max_tables, LJ_BW_DEC), s, a);
    mark(p, "lj_decode_kernel<multi>");
  }
}


} // namespace

void KernelTimer::begin(hipStream_t s) {
  stream = s;
  n = 0;
  if (created == 0 && hipEventCreate(&ev[0]) == hipSuccess)
    created = 1;
  if (created)
    (void)hipEventRecord(ev[0], s);
}

void KernelTimer::mark(const char* kernel) {
  if (n >= MAX || created == 0)
    return;
  if (created <= n + 1) {
    if (hipEventCreate(&ev[n + 1]) != hipSuccess)
      return;
    created = n + 2;
  }
  (void)hipEventRecord(ev[n + 1], stream);
  name[n++] = kernel;
}

int ljpeg_plan_create(rsx_ctx* ctx, const std::vector<LJpegJobIn>& jobs,
                      LJpegPlan** out) {
  auto p = std::make_unique<LJpegPlan>();
  p->ctx = ctx;
  p->n_jobs = int(jobs.size());
  p->job_status.assign(jobs.size(), RSX_OK);
  p->job_first_stream.assign(jobs.size(), -1);
  p->job_n_streams.assign(jobs.size(), 0);
  std::vector<DeviceHuffTable> tables;
  std::vector<Cr2Strip> strips;
  std::vector<uint32_t> nk_tables, nk_rowpow;
  std::map<uint32_t, uint32_t> nk_colpow_of; // row length -> its column powers in nk_rowpow
  for (size_t i = 0; i < jobs.size(); ++i) {
    const LJpegJobIn& J = jobs[i];
    int st = J.status;
    // BitStreamerJPEG needs >= 8 bytes, BitStreamerMSB >= 4 (BitStreamer.h:58-59)
    if (st == RSX_OK && J.geom.in_bytes < (J.geom.raw ? 4u : 8u))
      st = RSX_ERR_IO;
    if (st == RSX_OK && J.geom.in_bytes > 0xFFFFFFFFull)
      st = RSX_ERR_INVALID_ARG; // Buffer::size_type is uint32_t (io/Buffer.h:49)
    const StreamGeom& g = J.geom;
    // symbols of the entropy stage (a Hasselblad symbol is a pair of samples)
    const uint64_t needed = g.kind != 1
                                ? uint64_t(g.rows) * g.row_samples / (g.pair ? 2 : 1)
                                : g.strip_first_sample[g.n_strips];
    if (st == RSX_OK && needed >= 0xFFFFFFF0ull)
      st = RSX_ERR_UNSUPPORTED;
    // LJPEG / CR2 streams with 1, 2 or 4 interleaved components take the fused
    // decode + reconstruction; everything else (sRaw groups, 3 components, the
    // Nikon-type kinds, Hasselblad pairs) the legacy route through differences
    uint8_t direct_n = 0;
#ifndef RSX_NO_DIRECT
    if ((g.kind == 0 || g.kind == 1) && !g.raw && !g.las && !g.pair && !g.no_vertical &&
        g.period == g.n_comp && (g.n_comp == 1 || g.n_comp == 2 || g.n_comp == 4) &&
        J.explicit_n == 0)
      direct_n = uint8_t(g.n_comp);
#endif
    // The legacy route keeps a 2-byte difference per symbol, sized by the frame the
    // header declares, not by the input: a small tile that claims a huge frame is
    // refused (the forwarding hunks then leave it to the original CPU loop) instead of
    // asking the driver for gigabytes.
    if (st == RSX_OK && !direct_n && needed > (uint64_t(1) << 30))
      st = RSX_ERR_UNSUPPORTED;
    p->job_status[i] = st;
    if (st != RSX_OK)
      continue;
    if (g.kind == 0 && J.rows_per_restart_interval > 0 &&
        uint32_t(J.rows_per_restart_interval) < g.rows) {
      LJpegPlan::DriJob dj;
      dj.job = int(i);
      dj.in = J;
      dj.tables.assign(J.tables, J.tables + J.n_tables);
      dj.rows_per_ri = uint32_t(J.rows_per_restart_interval);
      dj.n_ri = (g.rows + dj.rows_per_ri - 1) / dj.rows_per_ri; // :277-278
      p->dri.push_back(std::move(dj));
      continue;
    }
    LjStreamDev S{};
    S.in_offset = g.in_offset;
    S.in_bytes = g.in_bytes;
    S.needed = needed;
    S.img_offset = g.img_offset;
    S.img_pitch = g.img_pitch_bytes;
    S.first_block = p->total_blocks;
    S.n_blocks = uint32_t((g.in_bytes + LJ_R - 1) / LJ_R);
    S.first_subseq = p->total_subseq;
    S.table_base = uint32_t(tables.size());
    S.n_tables = uint32_t(J.n_tables);
    S.period = g.period;
    S.n_comp = g.n_comp;
    std::memcpy(S.tab_of_phase, g.comp_of_phase, 8);
    std::memcpy(S.init_pred, g.init_pred, sizeof S.init_pred);
    std::memcpy(S.seed_pos, g.seed_pos, 4);
    S.raw = g.raw;
    S.start_bit = g.start_bit;
    S.las = g.las;
    S.pair = g.pair;
    S.no_vertical = g.no_vertical;
    S.direct = direct_n;
    S.diff_offset = direct_n ? LJ_NO_DIFFS : 0; // (legacy streams: assigned below)
    // the single-pass kernel: fused-path streams with one table whose MCU is a row of
    // its components (or CR2 strips)
#ifndef RSX_NO_FAST
    // (round 4: also TWO tables that alternate symbol by symbol -- one per component of a
    // two-component scan, the usual DNG; A B A B of four --: S.fast = 2, the kernel's
    // two-table instantiation, the table being part of every parse state)
    bool two_alternating = J.n_tables == 2 && (g.n_comp == 2 || g.n_comp == 4) &&
                           g.comp_of_phase[0] != g.comp_of_phase[1];
    for (uint32_t ph = 0; ph < g.n_comp && two_alternating; ++ph)
      two_alternating = g.comp_of_phase[ph] == g.comp_of_phase[ph & 1u];
#ifdef RSX_NO_FAST_MT
    two_alternating = false;
#endif
    // (round 6: ANY assignment of up to four tables to the components -- a table per
    // component of a three-component scan, what DNG writers emit for linear images; A B C D,
    // A B B, ... --: S.fast = 3, the kernel's table-per-phase instantiation, the phase
    // (symbol index mod N) being part of every parse state.  The table word of a workgroup
    // holds 16 bits of table base and 4 bits a phase.)
    bool per_phase = J.n_tables >= 2 && !two_alternating && g.n_comp >= 2 && g.n_comp <= 4 &&
                     g.period == g.n_comp && tables.size() + size_t(J.n_tables) < 0xFFFFu;
    for (uint32_t ph = 0; ph < g.n_comp && per_phase; ++ph)
      per_phase = g.comp_of_phase[ph] < J.n_tables && g.comp_of_phase[ph] < 16;
    // Tables that are NEARLY the same (the same code for most bit patterns: channels whose
    // statistics coincide) leave a parse no way to tell the phases apart from the bits, and they
    // differ just often enough that a parse in the wrong phase miscounts now and then: K0's chain
    // would settle in hardly any workgroup (tests/test_per_phase_model.py).  Such streams keep
    // the route they had; tables that differ in a length early in the canonical order share next
    // to nothing (measured: 0-13 % of the 11-bit patterns for unrelated tables, 74-99 % for
    // tables with two values swapped).
    if (per_phase && J.explicit_n == 0) {
      std::vector<DeviceHuffTable> tmp(size_t(J.n_tables));
      for (int t = 0; t < J.n_tables; ++t)
        build_device_table(J.tables[t], &tmp[size_t(t)], false);
      for (int x = 0; x < J.n_tables && per_phase; ++x)
        for (int y = x + 1; y < J.n_tables && per_phase; ++y) {
          uint32_t same = 0;
          for (uint32_t i = 0; i < uint32_t(LUT_SIZE); ++i)
            same += tmp[size_t(x)].lut[i] != 0 && tmp[size_t(x)].lut[i] == tmp[size_t(y)].lut[i];
          if (2u * same > uint32_t(LUT_SIZE))
            per_phase = false;
        }
    }
#ifdef RSX_NO_FAST_PT
    per_phase = false;
#endif
    S.fast = (direct_n && (J.n_tables == 1 || two_alternating || per_phase) &&
              (g.kind == 1 || (g.mcu_h == 1 && g.mcu_w == g.n_comp)) &&
              g.row_samples >= g.n_comp && tables.size() < 0xFFFFu)
                 ? (J.n_tables == 1 ? 1 : (two_alternating ? 2 : 3))
                 : 0;
    // Round 5: THREE interleaved components (MCU 3 x 1: linear DNG, LJpegDecompressor.cpp:
    // 102-105) with one table on the single-pass kernel's <3> instantiation -- the rotation
    // of the component sums by symbol counts mod 3 spelt out.  Such a stream is not one of
    // the fused multi-kernel path (direct stays 0: that path's kernels are for 1, 2 and 4
    // components), so a stream the kernel gives up on is redone by the legacy route, whose
    // difference scratch it keeps.
#ifndef RSX_NO_FAST3
    const bool fast3 = !direct_n && g.kind == 0 && !g.raw && !g.las && !g.pair &&
                       !g.no_vertical && g.n_comp == 3 && g.period == 3 && g.mcu_h == 1 &&
                       g.mcu_w == 3 && (J.n_tables == 1 || per_phase) && J.explicit_n == 0 &&
                       g.row_samples >= 3 && g.row_samples % 3 == 0;
    if (fast3)
      S.fast = J.n_tables == 1 ? 1 : 3;
    // Round 6: a stream whose RECONSTRUCTION the legacy kernels do -- a Nikon-type predictor with its
    // curve and dither, Pentax, SamsungV1's cousins with a canonical table, Canon sRaw groups -- and
    // that has one canonical table takes the single-pass kernel for its entropy half: the kernel
    // leaves the differences where lj_decode_kernel would have (its <1, 0, ., true> instantiation),
    // the reconstruction kernels follow in the same pass.  What it gives up on is redone by the
    // legacy route's own decode, like a three-component stream's.
    // ... and a Nikon-type stream -- NikonDecompressor without a split, PentaxDecompressor with at most
    // 15 bits -- has its PIXELS written by that kernel (fast_nk: the <2, 0, ., false, true>
    // instantiation; the vertical sums by row parity, curve and dither in the copy-out).  The kernel's
    // sums are mod 2^16: a value outside 0 .. 32767 (outside the sensor's bits for Pentax) hands the
    // stream to the legacy route, whose sums are the reference's ints.
    // (RSX_NO_FAST_NK / RSX_NO_FAST_DIFFS in the environment: the route one further back, for the
    // tests that compare the routes and for A/B timings)
    const bool env_no_nk = getenv("RSX_NO_FAST_NK") != nullptr;
    const bool env_no_diffs = getenv("RSX_NO_FAST_DIFFS") != nullptr;
    S.fast_nk = 0;
#ifndef RSX_NO_FAST_NK
    // (a table given as the reference's own (encLen, diffLen) pairs -- SamsungV1 -- is as good as a
    // canonical one while its codes fit the kernel's 10-bit LUT: every entry is a whole symbol)
    const bool lut10 = J.explicit_n == 0 || J.explicit_bits <= 10;
    if (!env_no_nk && !S.fast && !direct_n && g.kind == 2 && J.n_tables == 1 && lut10 && !g.las && !g.pair &&
        tables.size() < 0xFFFFu && needed >= 2 && g.row_samples >= 2 && g.row_samples % 2 == 0) {
      const NikonIn& N = J.nikon;
      bool ok = !N.sony && N.pup_in == nullptr && !(N.split > 0 && N.split < N.height) &&
                (!N.pentax || (N.range_bits >= 1 && N.range_bits <= 16)) &&
                (N.uncorrected || N.pentax || N.dither.size() == 32768);
      // (Pentax with 16 bits: a value from 32768 on could as well be a negative int -- the legacy
      // route's to tell)
      const int lim = N.pentax && N.range_bits < 15 ? (1 << N.range_bits) : 32768;
      for (int k = 0; k < 4 && ok; ++k)
        ok = N.p_up[k] >= 0 && N.p_up[k] < lim;
      if (ok) {
        S.fast = 1;
        S.fast_nk = 1;
        // (pUp by the parity of the STREAM row: stream row r is image row out_y + r)
        for (uint32_t h = 0; h < 2; ++h)
          for (uint32_t c = 0; c < 2; ++c)
            S.init_pred[2 * h + c] = uint16_t(N.p_up[2 * ((g.out_y + h) & 1u) + c]);
      }
    }
#endif
    S.fast_diffs = 0;
#ifndef RSX_NO_FAST_DIFFS
    if (!env_no_diffs && !S.fast && !direct_n && J.n_tables == 1 && lut10 && !g.las && !g.pair &&
        (g.kind == 2 || g.kind == 1 || g.kind == 0) && !(g.kind == 2 && J.nikon.sony) &&
        tables.size() < 0xFFFFu && needed >= 1) {
      S.fast = 1;
      S.fast_diffs = 1;
    }
#endif
    S.tab_period = 0;
    if (S.fast == 3) {
      // (the period of the assignment: A B A B over four components is two phases, not four --
      // phases the tables cannot tell apart would never fall into step)
      uint32_t tp = g.n_comp;
      for (uint32_t q = 1; q < g.n_comp; ++q) {
        bool ok = g.n_comp % q == 0;
        for (uint32_t ph = 0; ph < g.n_comp && ok; ++ph)
          ok = g.comp_of_phase[ph] == g.comp_of_phase[ph % q];
        if (ok) {
          tp = q;
          break;
        }
      }
      S.tab_period = uint8_t(tp);
    }
#endif
    if (S.fast) {
      // its workgroups stage all their samples in LDS at once: the allocation follows the
      // stream's symbols per workgroup (+ 15 % for local variation; a workgroup that
      // still does not fit sends the stream to the slow path)
      const uint64_t per_wg = (needed + S.n_blocks - 1) / (S.n_blocks ? S.n_blocks : 1);
      const uint32_t lds = ljpeg_fast_lds_for(per_wg + per_wg / 7 + 512);
      if (lds == 0)
        S.fast = 0; // (fewer than ~4 bits per symbol: the multi-kernel pipeline)
      else
        p->fast_lds = std::max(p->fast_lds, lds);
    }
#endif
    S.sync_lut11 = (J.explicit_n > 0 && J.explicit_bits > 10) ? 1 : 0;
    S.raw_limit = g.raw_limit;
    S.rows = g.rows;
    S.row_samples = g.row_samples;
    S.first_row = p->total_rows;
    S.kind = g.kind;
    S.mcu_w = g.mcu_w;
    S.mcu_h = g.mcu_h;
    S.out_x = g.out_x;
    S.out_y = g.out_y;
    S.keep_samples = g.keep_samples;
    if (g.kind == 0) {
      const uint32_t mcus = (g.keep_samples + g.mcu_w - 1) / g.mcu_w;
      S.scan_samples = std::min(g.row_samples, mcus * g.n_comp);
    } else {
      S.scan_samples = g.row_samples;
    }
    S.n_strips = g.n_strips;
    S.strip_base = uint32_t(strips.size());
    for (uint32_t k = 0; k < g.n_strips; ++k)
      strips.push_back({g.strip_x0[k], g.strip_w[k], g.strip_y0[k], g.strip_h[k],
                        g.strip_first_sample[k]});
    if (g.kind == 1)
      strips.push_back({0, 1, 0, 0, g.strip_first_sample[g.n_strips]});
    S.job = uint32_t(i);
    if (J.explicit_n > 0) {
      tables.emplace_back();
      build_device_table_explicit(J.explicit_enc_len, J.explicit_diff_len, J.explicit_n,
                                  &tables.back(), J.explicit_bits);
    } else {
      for (int t = 0; t < J.n_tables; ++t) {
        tables.emplace_back();
        rsx_huff_table ht = J.tables[t];
        // Hasselblad's getBits(16) reads 16 bits (HasselbladDecompressor.cpp:60-69);
        // in JPEG terms that is the "SSSS = 16 is followed by 16 bits" variant
        if (g.pair)
          ht.fix_dng_bug16 = 1;
        build_device_table(ht, &tables.back(), g.las != 0);
      }
    }
    NkStreamDev K{};
    if (g.kind == 2) {
      const NikonIn& N = J.nikon;
      std::memcpy(K.p_up, N.p_up, sizeof K.p_up);
      K.pup_in = N.pup_in;
      K.uncorrected = N.uncorrected ? 1u : 0u;
      K.pentax = N.pentax ? uint32_t(N.range_bits) : 0u;
      K.sony = N.sony ? 1u : 0u;
      K.seed_offset = N.seed_offset;
      K.table_off = uint32_t(nk_tables.size());
      if (!N.uncorrected)
        nk_tables.insert(nk_tables.end(), N.dither.begin(), N.dither.end());
      // 15700^(y * W) mod m for the stream's rows: the dither state at (y, 0)
      K.rowpow_off = uint32_t(nk_rowpow.size());
      K.colpow_off = 0;
      if (!N.uncorrected) {
        const uint64_t m = 15700ull * 65536 - 1;
        auto powmod = [&](uint64_t e) {
          uint64_t r = 1, b = 15700;
          for (; e; e >>= 1, b = b * b % m)
            if (e & 1)
              r = r * b % m;
          return r;
        };
        const uint64_t step = powmod(g.row_samples);
        uint64_t cur = powmod(uint64_t(g.out_y) * g.row_samples);
        for (uint32_t r = 0; r < g.rows; ++r, cur = cur * step % m)
          nk_rowpow.push_back(uint32_t(cur));
        if (S.fast_nk) {
          // 15700^x mod m for the columns (one list per row length of the plan)
          auto it = nk_colpow_of.find(g.row_samples);
          if (it == nk_colpow_of.end()) {
            it = nk_colpow_of.emplace(g.row_samples, uint32_t(nk_rowpow.size())).first;
            uint64_t c = 1;
            for (uint32_t x = 0; x < g.row_samples; ++x, c = c * 15700 % m)
              nk_rowpow.push_back(uint32_t(c));
          }
          K.colpow_off = it->second;
        }
      }
      p->any_nikon = true;
      if (N.split > 0 && N.pup_in == nullptr && g.out_y == 0 &&
          uint32_t(N.split) == g.rows && N.split < N.height) {
        LJpegPlan::NkSplit sp;
        sp.job = int(i);
        sp.stream = int(p->streams.size());
        sp.in = J;
        sp.table = N.table_after_split;
        p->nk_split.push_back(std::move(sp));
      }
    }
    p->nk.push_back(K);
    // single-table streams ignore tab_of_phase; multi-table ones index tabs[]
    const bool multi = J.n_tables > 1;
    p->any_multi |= multi;
    p->any_pair |= g.pair != 0;
    p->any_lut11 |= S.sync_lut11 != 0;
    if (!g.pair)
      p->sync_present[(multi || S.sync_lut11) ? 1 : 0][S.direct] = true;
    LJpegPlan::Classes& cls = S.direct ? p->fallback : p->legacy;
    cls.plain |= !multi && !g.las && !g.pair;
    cls.las |= g.las != 0;
    cls.pair |= g.pair != 0;
    cls.multi |= multi;
    cls.nikon |= g.kind == 2;
    cls.sony |= g.kind == 2 && J.nikon.sony;
    if (g.kind != 2)
      cls.comp[g.period == g.n_comp ? g.n_comp : (g.period == 4 ? 5u : 6u)] = true;
    if (S.fast) {
      p->any_fast = true;
      p->any_fast_mt |= S.fast == 2;
      p->any_fast_pt |= S.fast == 3;
      if (S.fast == 3)
        p->pt_np = std::max(p->pt_np, uint32_t(S.tab_period));
      if (S.fast_diffs)
        p->any_fast_diffs = true;
      else if (S.fast_nk)
        p->any_fast_nk = true;
      else
        p->fast_present[S.fast - 1][S.direct ? S.direct : g.n_comp] = true;
    } else {
      p->any_pipeline = true;
    }
    if (S.direct) {
      p->any_direct = true;
      p->direct_present[multi ? 1 : 0][S.direct] = true;
    } else {
      // (a legacy-route stream the single-pass kernel takes first -- 3 components -- needs
      // the route's launches only in the pass that redoes what that kernel gave up on)
      if (S.fast)
        p->any_fast_legacy = true;
      else
        p->any_legacy = true;
      // stream-ordered int16 scratch of the legacy route
      S.diff_offset = p->total_diffs;
      p->total_diffs += ((g.pair ? 2 * needed : needed) + 7 + 8) & ~uint64_t(7);
    }
    p->max_tables = std::max(p->max_tables, J.n_tables);
    p->job_first_stream[i] = int(p->streams.size());
    p->job_n_streams[i] = 1;
    p->total_blocks += S.n_blocks;
    p->total_subseq += S.n_blocks * LJ_OWN;
    p->total_rows += S.rows;
    p->streams.push_back(S);
  }
  // A plan with table-per-phase streams runs K0's table-per-phase instantiation, which knows
  // one-table streams and those: its two-table streams become table-per-phase streams too
  // (A B over 2 or 4 components is a pattern like any other; the kernel's <N, 2> instantiations).
  if (p->any_fast_pt && p->any_fast_mt) {
    for (LjStreamDev& S : p->streams)
      if (S.fast == 2) {
        S.fast = 3;
        S.tab_period = 2;
        p->fast_present[2][S.direct ? S.direct : S.n_comp] = true;
      }
    p->any_fast_mt = false;
    for (int n = 0; n < 5; ++n)
      p->fast_present[1][n] = false;
  }
  if (!p->streams.empty()) {
    std::vector<uint32_t> block_stream(p->total_blocks);
    for (size_t s = 0; s < p->streams.size(); ++s)
      for (uint32_t b = 0; b < p->streams[s].n_blocks; ++b)
        block_stream[p->streams[s].first_block + b] = uint32_t(s);
    // TabLds and DeviceHuffTable share their prefix; tables are uploaded in
    // the LDS layout (16-byte sized records)
    std::vector<TabLds> tl(tables.size());
    for (size_t t = 0; t < tables.size(); ++t) {
      const DeviceHuffTable& dt = tables[t];
      TabLds& o = tl[t];
      std::memset(&o, 0, sizeof(TabLds));
      std::memcpy(o.max_code, dt.max_code, sizeof o.max_code);
      std::memcpy(o.val_offset, dt.val_offset, sizeof o.val_offset);
      std::memcpy(o.values, dt.values, sizeof o.values);
      o.max_len = dt.max_len;
      o.fix16 = dt.fix16;
      o.zero_sym_bits = dt.zero_sym_bits;
      o.las = dt.las;
      for (uint32_t i = 0; i < uint32_t(LUT_SIZE); ++i) {
        const uint32_t e = dt.lut[i];
        uint32_t hi = 0;
#if RSX_LUT_DIFF
        // the symbol's difference, if code and difference bits lie inside the index
        const uint32_t cl = e & 31u, ssss = (e >> 5) & 31u, total = e >> 10;
        if (e != 0 && !dt.las && total <= uint32_t(LUT_BITS)) {
          if (ssss == 16u) {
            hi = 0x8000u;
          } else if (ssss != 0u) {
            const uint32_t v = (i >> (uint32_t(LUT_BITS) - total)) & ((1u << ssss) - 1u);
            hi = (v >= (1u << (ssss - 1)) ? v : v + 1u - (1u << ssss)) & 0xFFFFu;
          }
          (void)cl;
        }
#endif
        o.lut[i] = LutEntry(e | (hi << 16));
      }
    }
    auto up = [&](DeviceBuffer& b, const void* src, size_t n) -> int {
      if (int st = b.ensure(n ? n : 16))
        return st;
      if (n && hipMemcpy(b.ptr, src, n, hipMemcpyHostToDevice) != hipSuccess)
        return RSX_ERR_DEVICE;
      return RSX_OK;
    };
    int st = RSX_OK;
    if ((st = up(p->d_streams, p->streams.data(),
                 p->streams.size() * sizeof(LjStreamDev))) ||
        (st = up(p->d_tables, tl.data(), tl.size() * sizeof(TabLds))) ||
        (st = up(p->d_block_stream, block_stream.data(),
                 block_stream.size() * sizeof(uint32_t))) ||
        (st = up(p->d_strips, strips.data(), strips.size() * sizeof(Cr2Strip))) ||
        ((p->h_strips = strips), false))
      return st;
    if (p->any_nikon &&
        ((st = up(p->d_nk, p->nk.data(), p->nk.size() * sizeof(NkStreamDev))) ||
         (st = up(p->d_nk_tables, nk_tables.data(), nk_tables.size() * 4)) ||
         (st = up(p->d_nk_rowpow, nk_rowpow.data(), nk_rowpow.size() * 4)) ||
         (st = p->d_nk_pup.ensure(p->streams.size() * 16))))
      return st;
    if ((st = p->d_sub_state.ensure(size_t(p->total_subseq) * 4 + 16)) ||
        (st = p->d_block_start.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_flags.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_tf.ensure(size_t(p->total_blocks) * 64)) ||
        (st = p->d_sub_start.ensure(size_t(p->total_subseq) * 2 + 16)) ||
        (st = p->d_block_exit.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_sum.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_base.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_drops.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_block_drop_base.ensure(size_t(p->total_blocks) * 4)) ||
        (st = p->d_unstuffed.ensure(size_t(p->total_blocks) * LJ_IMG_U4 * 16)) ||
        (st = p->d_results.ensure(2 * p->streams.size() * sizeof(LjResult))) ||
        (st = p->d_diffs.ensure(size_t(p->total_diffs) * 2 + 64)) ||
        (st = p->d_vseed.ensure(size_t(p->total_rows) * 8 + 16)))
      return st;
    if (p->any_fast) {
      std::vector<uint2> ft(tl.size() * 1024);
      std::vector<uint32_t> fz(tl.size());
      for (size_t t = 0; t < tl.size(); ++t)
        ljpeg_build_fast_table(tl[t], ft.data() + t * 1024, &fz[t]);
      if ((st = up(p->d_fast_z, fz.data(), fz.size() * 4)))
        return st;
      // ticket order of the single-pass launches: round robin over the streams
      std::vector<uint4> order;
      order.reserve(p->total_blocks);
      uint32_t max_blocks = 0;
      for (const LjStreamDev& S : p->streams)
        max_blocks = std::max(max_blocks, S.n_blocks);
      // Since round 5 a workgroup's ticket is its block index (the dispatcher starts a 1-D
      // grid's workgroups in order; every wait in the kernel is bounded), and block t runs on
      // XCD t % 8.  With a multiple of 8 streams in plain round robin a stream would live on
      // ONE XCD -- all its image loads, pixel stores and look-back polls through one L2
      // (measured on cfg 3: the kernel +4-10 %) -- so the streams' turn rotates from group to
      // group then; other counts spread a stream over 8 / gcd(streams, 8) XCDs by themselves.
      const size_t ns = p->streams.size();
#if defined(RSX_LF_ROTATE) && RSX_LF_ROTATE == 0
      p->fast_rotate = 0;
#elif defined(RSX_LF_ROTATE) && RSX_LF_ROTATE == 2
      p->fast_rotate = 1;
#else
      p->fast_rotate = (ns % 8 == 0) ? 1u : 0u;
#endif
      p->fast_uniform_nb = max_blocks;
      for (const LjStreamDev& S : p->streams)
        if (S.n_blocks != max_blocks)
          p->fast_uniform_nb = 0;
#ifdef RSX_LF_NO_UNIFORM
      p->fast_uniform_nb = 0;
#endif
      for (uint32_t k = 0; k < max_blocks; ++k)
        for (size_t i0 = 0; i0 < ns; ++i0) {
          const size_t si = p->fast_rotate ? (i0 + k) % ns : i0;
          if (k < p->streams[si].n_blocks)
            // (.z: the stream's first table | its tables by phase: lf_table_word)
            order.push_back(make_uint4(p->streams[si].first_block + k, uint32_t(si),
                                       lf_table_word(p->streams[si].table_base,
                                                     p->streams[si].tab_of_phase[0],
                                                     p->streams[si].tab_of_phase[1],
                                                     p->streams[si].tab_of_phase[2],
                                                     p->streams[si].tab_of_phase[3]),
                                       p->streams[si].first_block));
        }
      if ((st = up(p->d_fast_order, order.data(), order.size() * sizeof(uint4))))
        return st;
      if (p->any_fast_pt) {
        std::vector<uint16_t> ft16(tl.size() * 1024);
        for (size_t t = 0; t < tl.size(); ++t)
          ljpeg_build_fast_table16(ft.data() + t * 1024, ft16.data() + t * 1024);
        if ((st = up(p->d_fast_tabs16, ft16.data(), ft16.size() * sizeof(uint16_t))))
          return st;
      }
      if ((st = up(p->d_fast_tabs, ft.data(), ft.size() * sizeof(uint2))) ||
          (st = p->d_lb.ensure(size_t(p->total_blocks) * LF_LB_WORDS * 8)) ||
          (st = p->d_k0w.ensure(size_t(p->total_blocks + 1) * 8)) ||
          (st = p->d_k0e.ensure(size_t(p->total_blocks + 1) * 4)) ||
          (st = p->d_block_base0.ensure(size_t(p->total_blocks) * 4)) ||
          (st = p->d_tickets.ensure(LF_TICKET_WORDS * 4)))
        return st;
      if ((st = p->d_fast_level.ensure(256)))
        return st;
      RSX_HIP_CHECK(ctx, hipMemset(p->d_tickets.ptr, 0, LF_TICKET_WORDS * 4));
      RSX_HIP_CHECK(ctx, hipMemset(p->d_k0w.ptr, 0, size_t(p->total_blocks + 1) * 8));
      RSX_HIP_CHECK(ctx, hipMemset(p->d_k0e.ptr, 0, size_t(p->total_blocks + 1) * 4));
      if (p->any_fast_pt) {
        if ((st = p->d_k0p.ensure(2 * size_t(p->total_blocks + 1) * 4)))
          return st;
        RSX_HIP_CHECK(ctx, hipMemset(p->d_k0p.ptr, 0, 2 * size_t(p->total_blocks + 1) * 4));
      }
      RSX_HIP_CHECK(ctx, hipMemset(p->d_block_base0.ptr, 0, size_t(p->total_blocks) * 4));
      RSX_HIP_CHECK(ctx, hipMemset(p->d_fast_level.ptr, 0, 256));
      // hipMemset of device memory is ASYNCHRONOUS with respect to the host on this runtime (ROCm 7.2:
      // scripts/repro/memset_async.hip -- the call returns in 2 us, and a hipStreamNonBlocking stream
      // reads the old bytes for as long as the null stream is busy), and the plan's kernels run on
      // such a stream: without this wait the six memsets above raced with the plan's first run
      // whenever other host threads kept the null stream busy -- K0's words, the hand-over words or
      // the LDS level zeroed under the kernels' feet; once in ~17 000 calls from six threads a whole
      // small tile came back wrong with status OK (profiles/r06/host_path_defect/README.md).
#ifndef RSX_NO_CREATE_SYNC // (experiment: the behaviour until round 6, for scripts/exp_null_stream_stress.py)
      RSX_HIP_CHECK(ctx, hipStreamSynchronize(nullptr));
#endif
#ifdef RSX_EXPERIMENT
      if (getenv("RSX_DEBUG")) {
        if ((st = p->d_dbg.ensure(size_t(p->total_blocks) * 32 * 8)))
          return st;
        (void)hipMemset(p->d_dbg.ptr, 0, size_t(p->total_blocks) * 32 * 8);
        (void)hipStreamSynchronize(nullptr); // (asynchronous with respect to the host: see above)
      }
#endif
    }
    if (p->any_direct &&
        ((st = p->d_sub_sums.ensure(size_t(p->total_subseq) * 8 + 16)) ||
         (st = p->d_sub_first.ensure(size_t(p->total_subseq) * 4 + 16)) ||
         (st = p->d_sub_psum.ensure(size_t(p->total_subseq) * 8 + 16)) ||
         (st = p->d_block_psum.ensure(size_t(p->total_blocks) * 8)) ||
         (st = p->d_block_pbase.ensure(size_t(p->total_blocks) * 8)) ||
         (st = p->d_row_edge.ensure(size_t(p->total_rows) * 16 + 16))))
      return st;
    // (the single-pass kernel leaves its workgroups' difference sums there whatever route
    // its streams fall back to: 3-component streams are not fused-path streams)
    if (p->any_fast && !p->d_block_psum.ptr &&
        (st = p->d_block_psum.ensure(size_t(p->total_blocks) * 8)))
      return st;
    p->h_results.resize(p->streams.size());
  }
  *out = p.release();
  return RSX_OK;
}


namespace {

// the legacy route for the streams of class set `c`: decode into differences, tail,
// seeds + row scans
void launch_legacy(LJpegPlan* p, const LJpegPlan::Classes& c, const LjArgs& a,
                   hipStream_t s) {
  const uint32_t n_streams = uint32_t(p->streams.size());
  launch_decode(p, c, a, s);
  hipLaunchKernelGGL(lj_tail_kernel, dim3(n_streams), dim3(64), 0, s, a);
  mark(p, "lj_tail_kernel");
  ReconLaunch rl;
  rl.n_streams = n_streams;
  rl.total_rows = p->total_rows;
  std::copy(c.comp, c.comp + 7, rl.comp_present);
  rl.any_nikon = c.nikon;
  rl.any_sony = c.sony;
  ljpeg_launch_reconstruct(a, rl, s);
  mark(p, "legacy reconstruction (K5 + K6)");
}

// K1, the stitch passes and the chain of periodic workgroups
void launch_synchronisation(LJpegPlan* p, const LjArgs& a, hipStream_t s) {
  const uint32_t n_streams = uint32_t(p->streams.size());
  launch_sync<false>(p, a, s);
  for (int r = 0; r < p->stitch_rounds; ++r) {
    launch_sync<true>(p, a, s);
    if (r + 1 < p->stitch_rounds) {
      hipLaunchKernelGGL(lj_pchain_kernel, dim3(n_streams), dim3(64), 0, s, a);
      mark(p, "lj_pchain_kernel");
    }
  }
}

// everything after synchronisation and the scan: decode, reconstruct, consumed
// (pipeline: some stream of this launch takes the multi-kernel pipeline)
// (legacy, bits: 1 the plan's legacy-route streams, first pass; 2 those the single-pass kernel
// takes first and gave up on, second pass; 0 none)
int launch_tail(LJpegPlan* p, const LjArgs& a, hipStream_t s, bool pipeline = true,
                int legacy = 1) {
  rsx_ctx* ctx = p->ctx;
  const uint32_t n_streams = uint32_t(p->streams.size());
  if (p->any_direct && pipeline) {
    DirectLaunch dl;
    dl.n_streams = n_streams;
    dl.total_blocks = p->total_blocks;
    dl.total_rows = p->total_rows;
    dl.max_tables = p->max_tables;
    std::memcpy(dl.present, p->direct_present, sizeof dl.present);
    ljpeg_launch_direct(a, dl, s, p->timer);
  }
  if (((legacy & 1) && p->any_legacy) || ((legacy & 2) && p->any_fast_legacy)) {
    launch_legacy(p, p->legacy, a, s);
  } else if ((legacy & 1) && a.pass == 0 && p->any_fast_diffs) {
    // (the first pass of streams whose differences the single-pass kernel has just left: their
    // reconstruction -- the kernels skip the streams it gave up on, lj_recon_takes)
    ReconLaunch rl;
    rl.n_streams = n_streams;
    rl.total_rows = p->total_rows;
    std::copy(p->legacy.comp, p->legacy.comp + 7, rl.comp_present);
    rl.any_nikon = p->legacy.nikon;
    rl.any_sony = p->legacy.sony;
    ljpeg_launch_reconstruct(a, rl, s);
    mark(p, "legacy reconstruction (K5 + K6)");
  }
  if (!a.fuse_consumed) {
    hipLaunchKernelGGL(lj_consumed_kernel, dim3(n_streams), dim3(64), 0, s, a);
    mark(p, "lj_consumed_kernel");
  }
  RSX_HIP_CHECK(ctx, hipGetLastError());
  return RSX_OK;
}

// Second pass: the streams the single-pass kernel gave up on (FL_SLOW, set on the
// device) go through the multi-kernel pipeline; every other stream is left alone.
int launch_slow_pass(LJpegPlan* p, hipStream_t s) {
  LjArgs a = make_args(p, p->last_in, p->last_out);
  a.pass = 1;
  const uint32_t n_streams = uint32_t(p->streams.size());
  launch_synchronisation(p, a, s);
  hipLaunchKernelGGL(lj_scan_kernel<false>, dim3(n_streams), dim3(LJ_T), 0, s, a);
  mark(p, "lj_scan_kernel");
  p->slow_pass_launched = true;
  return launch_tail(p, a, s, true, 2);
}

} // namespace

namespace {

// Locate the RSTn markers of every restart-interval job, turn each interval
// into a stream of a child plan (fresh predictors, byte-aligned start:
// LJpegDecompressor.cpp:283-300) and run it.  One host round trip per run; the
// child plan is reused while the marker layout does not change.
int run_dri_host(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  const uint8_t* in_base = static_cast<const uint8_t*>(in_dev);
  std::vector<uint32_t> signature;
  // Marker scans of ALL jobs with restart intervals, then ONE round trip for their counts and
  // lists (round 3 synchronised once per job: four tiles, four round trips of ~50 us in a
  // 0.5 ms decode).
  const size_t nd = p->dri.size();
  std::vector<uint32_t> caps(nd), offs(nd), counts(nd, 0);
  size_t total_cap = 0;
  for (size_t d = 0; d < nd; ++d) {
    caps[d] = p->dri[d].n_ri + 1024;
    offs[d] = uint32_t(total_cap);
    total_cap += caps[d];
  }
  if (int st = p->d_marker_count.ensure(nd * 4 + 16))
    return st;
  if (int st = p->d_marker_list.ensure(total_cap * sizeof(uint2)))
    return st;
  RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_marker_count.ptr, 0, nd * 4, s));
  for (size_t d = 0; d < nd; ++d) {
    const auto& dj = p->dri[d];
    const uint64_t bytes = dj.in.geom.in_bytes;
    const uint32_t blocks = uint32_t((bytes + 4095) / 4096);
    hipLaunchKernelGGL(lj_marker_scan_kernel, dim3(blocks), dim3(256), 0, s,
                       in_base + dj.in.geom.in_offset, bytes,
                       static_cast<uint32_t*>(p->d_marker_count.ptr) + d,
                       static_cast<uint2*>(p->d_marker_list.ptr) + offs[d], caps[d]);
  }
  RSX_HIP_CHECK(ctx, hipGetLastError());
  std::vector<uint2> all_lists(total_cap);
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(counts.data(), p->d_marker_count.ptr, nd * 4,
                                    hipMemcpyDeviceToHost, s));
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(all_lists.data(), p->d_marker_list.ptr,
                                    total_cap * sizeof(uint2), hipMemcpyDeviceToHost, s));
  RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
  for (size_t d = 0; d < nd; ++d) {
    auto& dj = p->dri[d];
    dj.status = RSX_OK;
    const uint64_t bytes = dj.in.geom.in_bytes;
    uint32_t count = counts[d];
    std::vector<uint2> list;
    if (count <= caps[d]) {
      list.assign(all_lists.begin() + offs[d], all_lists.begin() + offs[d] + count);
    } else {
      // rare: lots of FFxx behind the scan -- this job again, with room for all of them
      const uint32_t cap = count;
      DeviceBuffer big;
      if (int st = big.ensure(size_t(cap) * sizeof(uint2)))
        return st;
      RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_marker_count.ptr, 0, 4, s));
      const uint32_t blocks = uint32_t((bytes + 4095) / 4096);
      hipLaunchKernelGGL(lj_marker_scan_kernel, dim3(blocks), dim3(256), 0, s,
                         in_base + dj.in.geom.in_offset, bytes,
                         static_cast<uint32_t*>(p->d_marker_count.ptr),
                         static_cast<uint2*>(big.ptr), cap);
      RSX_HIP_CHECK(ctx, hipGetLastError());
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(&count, p->d_marker_count.ptr, 4,
                                        hipMemcpyDeviceToHost, s));
      RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
      if (count > cap)
        count = cap;
      list.resize(count);
      if (count)
        RSX_HIP_CHECK(ctx, hipMemcpy(list.data(), big.ptr, size_t(count) * sizeof(uint2),
                                     hipMemcpyDeviceToHost));
      big.release();
    }
    std::sort(list.begin(), list.end(),
              [](const uint2& x, const uint2& y) { return x.x < y.x; });
    dj.markers.clear();
    dj.codes.clear();
    dj.starts.assign(1, 0u);
    for (uint32_t i = 0; i + 1 < dj.n_ri; ++i) {
      if (i >= list.size()) {
        dj.status = RSX_ERR_RESTART_MARKER; // "Jpeg marker not encountered"
        break;
      }
      dj.markers.push_back(list[i].x);
      dj.codes.push_back(uint8_t(list[i].y));
      dj.starts.push_back(list[i].x + 2);
    }
    signature.push_back(uint32_t(dj.status));
    signature.insert(signature.end(), dj.starts.begin(), dj.starts.end());
  }
  if (!p->child || signature != p->dri_signature) {
    if (p->child) {
      ljpeg_plan_destroy(p->child);
      p->child = nullptr;
    }
    std::vector<LJpegJobIn> jobs;
    p->child_owner.clear();
    for (size_t d = 0; d < p->dri.size(); ++d) {
      auto& dj = p->dri[d];
      if (dj.status != RSX_OK)
        continue;
      const StreamGeom& g = dj.in.geom;
      for (uint32_t i = 0; i < dj.n_ri; ++i) {
        LJpegJobIn J = dj.in;
        J.tables = dj.tables.data();
        J.rows_per_restart_interval = 0;
        const uint64_t start = dj.starts[i];
        // up to and including the closing marker (the reference hands every
        // interval the whole remaining buffer; >= 8 bytes: BitStreamer.h:58-59)
        uint64_t end = (i + 1 < dj.n_ri) ? uint64_t(dj.markers[i]) + 2 : g.in_bytes;
        end = std::min<uint64_t>(g.in_bytes, std::max<uint64_t>(end, start + 8));
        J.geom.in_offset = g.in_offset + start;
        J.geom.in_bytes = end - start;
        const uint32_t r0 = i * dj.rows_per_ri;
        J.geom.rows = std::min(dj.rows_per_ri, g.rows - r0);
        J.geom.out_y = g.out_y + g.mcu_h * r0;
        jobs.push_back(J);
        p->child_owner.emplace_back(int(d), int(i));
      }
    }
    if (!jobs.empty())
      if (int st = ljpeg_plan_create(ctx, jobs, &p->child))
        return st;
    p->dri_signature = signature;
  }
  if (p->child)
    return ljpeg_plan_run(p->child, in_dev, out_dev, s, nullptr);
  return RSX_OK;
}

// The same without the round trip: marker scans, sort, layout of the (once-made) child plan
// on the device, its run -- all in stream order.  RSX_INTERNAL_RETRY: not for this plan.
int run_dri_device(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  const uint8_t* in_base = static_cast<const uint8_t*>(in_dev);
  const size_t nd = p->dri.size();
  if (!p->child_dev) {
    // the child plan from the geometry alone: interval i of a job gets an equal share of
    // the blocks any marker placement can need (its real place and size come from the
    // layout kernel in every run)
    std::vector<LJpegJobIn> jobs;
    p->child_dev_owner.clear();
    p->dri_dev.assign(nd, DriJobDev{});
    size_t total_cap = 0, total_sorted = 0;
    for (size_t d = 0; d < nd; ++d) {
      auto& dj = p->dri[d];
      const StreamGeom& g = dj.in.geom;
      if (g.in_bytes < 16 || dj.n_ri < 2)
        return RSX_INTERNAL_RETRY;
      DriJobDev& D = p->dri_dev[d];
      D.in_offset = g.in_offset;
      D.in_bytes = g.in_bytes;
      D.n_ri = dj.n_ri;
      D.first_stream = uint32_t(jobs.size());
      D.list_off = uint32_t(total_cap);
      D.cap = dj.n_ri + 1024;
      D.sorted_off = uint32_t(total_sorted);
      total_cap += D.cap;
      total_sorted += dj.n_ri;
      // (intervals overlap by their closing marker and are at least 8 bytes long)
      const uint64_t bmax = (g.in_bytes + 10ull * dj.n_ri) / LJ_R + dj.n_ri + 2;
      const uint64_t share = (bmax + dj.n_ri - 1) / dj.n_ri;
      for (uint32_t i = 0; i < dj.n_ri; ++i) {
        LJpegJobIn J = dj.in;
        J.tables = dj.tables.data();
        J.rows_per_restart_interval = 0;
        J.geom.in_offset = g.in_offset;
        J.geom.in_bytes = share * LJ_R; // (a placeholder: sizes the plan's arrays)
        const uint32_t r0 = i * dj.rows_per_ri;
        J.geom.rows = std::min(dj.rows_per_ri, g.rows - r0);
        J.geom.out_y = g.out_y + g.mcu_h * r0;
        jobs.push_back(J);
        p->child_dev_owner.emplace_back(int(d), int(i));
      }
    }
    LJpegPlan* c = nullptr;
    if (int st = ljpeg_plan_create(ctx, jobs, &c))
      return st == RSX_ERR_NOMEM || st == RSX_ERR_DEVICE ? st : RSX_INTERNAL_RETRY;
    // only plans whose streams ALL take the single-pass kernel (its kernels and K0 are the
    // ones that skip blocks without a stream)
    bool ok = c->any_fast && !c->any_pipeline && !c->any_legacy && !c->any_fast_legacy &&
              c->dri.empty() && c->nk_split.empty() && c->streams.size() == jobs.size();
    for (int st : c->job_status)
      ok = ok && st == RSX_OK;
    if (!ok) {
      ljpeg_plan_destroy(c);
      return RSX_INTERNAL_RETRY;
    }
    c->dev_layout = true;
    c->fast_uniform_nb = 0;
    // (the child becomes the plan's only once every array the launches below use exists and
    // the job records are on the device: a later run must not find a child without them)
    int st = p->d_dri_jobs.ensure(nd * sizeof(DriJobDev));
    if (st == RSX_OK)
      st = p->d_dri_sorted.ensure(total_sorted * sizeof(uint2) + 16);
    if (st == RSX_OK)
      st = p->d_dri_status.ensure((nd + 1) * 4 + 16);
    if (st == RSX_OK)
      st = p->d_marker_count.ensure(nd * 4 + 16);
    if (st == RSX_OK)
      st = p->d_marker_list.ensure(total_cap * sizeof(uint2));
    if (st == RSX_OK &&
        hipMemcpy(p->d_dri_jobs.ptr, p->dri_dev.data(), nd * sizeof(DriJobDev),
                  hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      st = RSX_ERR_DEVICE;
    }
    if (st != RSX_OK) {
      ljpeg_plan_destroy(c);
      return st;
    }
    p->child_dev = c;
  }
  LJpegPlan* c = p->child_dev;
  RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_marker_count.ptr, 0, nd * 4, s));
  RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_dri_status.ptr, 0, (nd + 1) * 4, s));
  uint32_t max_cap = 0;
  uint64_t max_bytes = 0;
  for (size_t d = 0; d < nd; ++d) {
    max_bytes = std::max<uint64_t>(max_bytes, p->dri[d].in.geom.in_bytes);
    max_cap = std::max(max_cap, p->dri_dev[d].cap);
  }
  hipLaunchKernelGGL(lj_dri_scan_kernel, dim3(uint32_t((max_bytes + 15 + 4095) / 4096), uint32_t(nd)),
                     dim3(256), 0, s, in_base, static_cast<const DriJobDev*>(p->d_dri_jobs.ptr),
                     static_cast<uint32_t*>(p->d_marker_count.ptr),
                     static_cast<uint2*>(p->d_marker_list.ptr));
  hipLaunchKernelGGL(lj_dri_sort_kernel, dim3((max_cap + 255) / 256, uint32_t(nd)), dim3(256), 0,
                     s, static_cast<const DriJobDev*>(p->d_dri_jobs.ptr),
                     static_cast<const uint32_t*>(p->d_marker_count.ptr),
                     static_cast<const uint2*>(p->d_marker_list.ptr),
                     static_cast<uint2*>(p->d_dri_sorted.ptr));
  hipLaunchKernelGGL(lj_dri_layout_kernel, dim3(1), dim3(1024), 0, s,
                     static_cast<LjStreamDev*>(c->d_streams.ptr), uint32_t(c->streams.size()),
                     static_cast<uint32_t*>(c->d_block_stream.ptr),
                     static_cast<uint4*>(c->d_fast_order.ptr), c->total_blocks,
                     static_cast<const DriJobDev*>(p->d_dri_jobs.ptr), uint32_t(nd),
                     static_cast<const uint32_t*>(p->d_marker_count.ptr),
                     static_cast<const uint2*>(p->d_dri_sorted.ptr),
                     static_cast<uint32_t*>(p->d_dri_status.ptr));
  // (the child's K0, single-pass kernel and scan read what the layout kernel has just written
  // through the scalar cache: the child is a dev_layout plan, its launches are the
  // instantiations that invalidate that cache in every wavefront -- lj_fresh_scalars)
  RSX_HIP_CHECK(ctx, hipGetLastError());
  mark(p, "lj_dri_scan + lj_dri_sort + lj_dri_layout");
  return ljpeg_plan_run_(c, in_dev, out_dev, s, p->timer, true);
}

int run_dri(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  p->dri_ran_dev = false;
#ifndef RSX_DRI_HOST_ONLY
  if (!p->child_dev_failed) {
    const int st = run_dri_device(p, in_dev, out_dev, s);
    if (st == RSX_OK) {
      p->dri_ran_dev = true;
      return RSX_OK;
    }
    if (st != RSX_INTERNAL_RETRY)
      return st;
    p->child_dev_failed = true;
  }
#endif
  return run_dri_host(p, in_dev, out_dev, s);
}

} // namespace

namespace {

int converge(LJpegPlan* p, hipStream_t s);

// NikonDecompressor::decompress with a split (NikonDecompressor.cpp:555-559):
// the rows from `split` on are decoded with the next table by the same bit
// reader, i.e. they start at the bit where the first part ended.  One host
// round trip per run; the child plan is reused while those bits do not move.
int run_nikon_split(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  if (int st = converge(p, s))
    return st;
  std::vector<uint64_t> signature;
  for (auto& sp : p->nk_split) {
    const LjResult& R = p->h_results[sp.stream];
    const LjStreamDev& S = p->streams[sp.stream];
    sp.status = RSX_OK;
    if (R.status != 0 || uint64_t(R.avail_lo) < S.needed) {
      signature.push_back(~uint64_t(0)); // the first part failed; reported by its stream
      continue;
    }
    signature.push_back((uint64_t(R.end_hi) << 32) | R.end_lo);
  }
  if (!p->nk_child || signature != p->nk_signature) {
    if (p->nk_child) {
      ljpeg_plan_destroy(p->nk_child);
      p->nk_child = nullptr;
    }
    std::vector<LJpegJobIn> jobs;
    p->nk_child_owner.clear();
    for (size_t k = 0; k < p->nk_split.size(); ++k) {
      auto& sp = p->nk_split[k];
      if (signature[k] == ~uint64_t(0))
        continue;
      const StreamGeom& g = sp.in.geom;
      const uint64_t end = signature[k], off = end / 8;
      const uint64_t rest =
          uint64_t(sp.in.nikon.height - sp.in.nikon.split) * g.row_samples;
      if (off + 4 > g.in_bytes) {
        // fewer than 4 bytes left: the reference keeps reading zeros for at
        // most 8 + 3 more bytes, far less than `rest` symbols need
        sp.status = rest > 48 ? RSX_ERR_INPUT_OVERFLOW : RSX_ERR_UNSUPPORTED;
        continue;
      }
      LJpegJobIn J = sp.in;
      J.geom.in_offset = g.in_offset + off;
      J.geom.in_bytes = g.in_bytes - off;
      J.geom.start_bit = uint8_t(end % 8);
      // the position budget belongs to the whole input (BitStreamer.h:125-127)
      J.geom.raw_limit = 32 * ((g.in_bytes + 8) / 4) - 8 * off + 1;
      J.geom.rows = uint32_t(sp.in.nikon.height - sp.in.nikon.split);
      J.geom.out_y = uint32_t(sp.in.nikon.split);
      J.geom.las = 1;
      J.tables = &sp.table;
      J.n_tables = 1;
      J.nikon.split = 0;
      J.nikon.pup_in = static_cast<const int32_t*>(p->d_nk_pup.ptr) + 4 * sp.stream;
      jobs.push_back(J);
      p->nk_child_owner.push_back(int(k));
    }
    if (!jobs.empty())
      if (int st = ljpeg_plan_create(ctx, jobs, &p->nk_child))
        return st;
    p->nk_signature = signature;
  }
  if (p->nk_child)
    return ljpeg_plan_run(p->nk_child, in_dev, out_dev, s, nullptr);
  return RSX_OK;
}

} // namespace

int ljpeg_plan_run(LJpegPlan* p, const void* in_dev, void* out_dev,
                   hipStream_t s, KernelTimer* timer) {
  return ljpeg_plan_run_(p, in_dev, out_dev, s, timer, false);
}
// A run in three parts -- begin (bookkeeping, the results' set), the BLOCKS of K0 and the
// single-pass kernel, end (pipeline, scan, tail, slow pass) --: ljpeg_plan_run_ queues all blocks
// at once; a host-pointer call of one large stream queues them chunk by chunk, each behind the
// upload of its bytes, and fetches the pixels a chunk completes while the next one decodes
// (rsx_api.hip, ljpeg_family_host; round 6).
static int ljpeg_plan_run_begin_(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  ++p->run_count;
  // demoted streams get the single-pass kernel back after DEMOTION_RUNS runs
  if (!p->demoted_at.empty()) {
    bool back = false;
    for (size_t k = 0; k < p->streams.size(); ++k)
      if (p->demoted_at[k] != 0 && p->run_count - p->demoted_at[k] >= LJpegPlan::DEMOTION_RUNS) {
        p->streams[k].fast = p->demoted_fast[k];
        p->demoted_at[k] = 0;
        p->slow_strikes[k] = 0;
        back = true;
      }
    if (back) {
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(p->d_streams.ptr, p->streams.data(),
                                        p->streams.size() * sizeof(LjStreamDev),
                                        hipMemcpyHostToDevice, s));
      // (K0's hand-over words carry one parity bit as their epoch, and a demoted stream does
      // not write its own: after an even number of runs the words of the run it was demoted
      // in would pass for this run's.  Start the re-promoted streams from clean ones.)
      if (p->d_k0e.ptr)
        RSX_HIP_CHECK(ctx, hipMemsetAsync(p->d_k0e.ptr, 0, size_t(p->total_blocks + 1) * 4, s));
      RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
      p->any_fast = true;
      // the demotion decays completely: what the first pass launches follows the streams
      p->any_pipeline = false;
      p->any_legacy = p->any_fast_legacy = false;
      for (const LjStreamDev& S : p->streams) {
        p->any_pipeline |= S.fast == 0;
        p->any_legacy |= !S.fast && !S.direct;
        p->any_fast_legacy |= S.fast && !S.direct;
      }
      p->expect_slow = false;
    }
  }
  LjArgs a = make_args(p, in_dev, out_dev);
  a.fuse_consumed = (p->any_fast && !p->any_pipeline && !p->any_legacy) ? 1u : 0u;
#ifndef RSX_NO_FIRST_RUN_INV
  // A plan's FIRST run: its kernels read the stream records, the block map and the ticket
  // order through the scalar cache, nothing invalidates a CU's scalar cache between two
  // kernels of a stream, and the arrays of a new plan tend to lie where those of the plan
  // destroyed just before it lay (a host-pointer call with new geometry: the lane's cached
  // plan is replaced).  5 us once per plan.
  if (p->run_count == 1 && !p->dev_layout)
    hipLaunchKernelGGL(lj_dcache_inv_kernel, dim3(4096), dim3(64), 0, s);
#endif
  // results: marker_pos = 0xFFFFFFFF, everything else 0
  for (auto& r : p->h_results) {
    std::memset(&r, 0, sizeof r);
    r.marker_pos = 0xFFFFFFFFu;
  }
  // (on the device: a copy of a few hundred bytes from pageable memory in front of every run
  // costs the stream more than a kernel of one wavefront)
  const uint32_t n_streams = uint32_t(p->streams.size());
#ifdef RSX_RESULTS_BY_COPY
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(results_of_run(p), p->h_results.data(),
                                    p->h_results.size() * sizeof(LjResult),
                                    hipMemcpyHostToDevice, s));
#else
  // (unless the run before this one has cleared them: K0 does that for its successor, on the
  // same stream)
  if (p->results_clean_for != p->run_count || p->results_clean_stream != s) {
    hipLaunchKernelGGL(lj_init_results_kernel, dim3((n_streams * uint32_t(sizeof(LjResult) / 4) + 255) / 256),
                       dim3(256), 0, s, results_of_run(p), n_streams);
    mark(p, "lj_init_results_kernel");
  }
#endif
  p->results_clean_for = p->run_count + 1;
  p->results_clean_stream = s;
  p->run_args = a;
  return RSX_OK;
}

// K0 and the single-pass kernel over the plan's blocks [blk0, blk1)
static int ljpeg_plan_run_blocks_(LJpegPlan* p, hipStream_t s, uint32_t blk0, uint32_t blk1) {
  rsx_ctx* ctx = p->ctx;
  if (blk1 <= blk0)
    return RSX_OK;
  LjArgs a = p->run_args;
  a.blk0 = blk0;
  a.blk_n = blk1 - blk0;
  // (plans laid out on the device: the instantiations whose wavefronts drop the scalar cache
  // first -- lj_fresh_scalars)
  if (p->any_fast_pt && p->dev_layout)
    hipLaunchKernelGGL((lj_unstuff_kernel<2, true>), dim3(a.blk_n), dim3(LJ_T),
                       lj_k0_lds_pt(p->pt_np), s, a);
  else if (p->any_fast_pt)
    hipLaunchKernelGGL((lj_unstuff_kernel<2, false>), dim3(a.blk_n), dim3(LJ_T),
                       lj_k0_lds_pt(p->pt_np), s, a);
  else if (p->any_fast_mt && p->dev_layout)
    hipLaunchKernelGGL((lj_unstuff_kernel<1, true>), dim3(a.blk_n), dim3(LJ_T),
                       LJ_K0_LDS_MT, s, a);
  else if (p->any_fast_mt)
    hipLaunchKernelGGL((lj_unstuff_kernel<1, false>), dim3(a.blk_n), dim3(LJ_T),
                       LJ_K0_LDS_MT, s, a);
  else if (p->dev_layout)
    hipLaunchKernelGGL((lj_unstuff_kernel<0, true>), dim3(a.blk_n), dim3(LJ_T),
                       LJ_K0_LDS, s, a);
  else
    hipLaunchKernelGGL((lj_unstuff_kernel<0, false>), dim3(a.blk_n), dim3(LJ_T),
                       LJ_K0_LDS, s, a);
  mark(p, "lj_unstuff_kernel");
  // the single-pass kernel for the streams it takes ...
  if (p->any_fast) {
    FastLaunch fl;
    fl.total_blocks = a.blk_n;
    std::memcpy(fl.present, p->fast_present, sizeof fl.present);
    fl.diffs = p->any_fast_diffs;
    fl.nk = p->any_fast_nk;
    // one single-pass launch of a context at a time (rsx_ctx::fast_mu)
    std::lock_guard<std::mutex> g(ctx->fast_mu);
    if (ctx->fast_ev_valid && ctx->fast_ev_stream != s)
      RSX_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->fast_ev, 0));
    ljpeg_launch_fast(a, fl, s, p->timer);
    if (!ctx->fast_ev)
      RSX_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->fast_ev, hipEventDisableTiming));
    RSX_HIP_CHECK(ctx, hipEventRecord(ctx->fast_ev, s));
    ctx->fast_ev_valid = true;
    ctx->fast_ev_stream = s;
  }
  return RSX_OK;
}

static int ljpeg_plan_run_end_(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  const LjArgs a = p->run_args;
  const uint32_t n_streams = uint32_t(p->streams.size());
  // ... the multi-kernel pipeline for the others
  if (p->any_pipeline)
    launch_synchronisation(p, a, s);
  if (p->dev_layout)
    hipLaunchKernelGGL(lj_scan_kernel<true>, dim3(n_streams), dim3(LJ_T), 0, s, a);
  else
    hipLaunchKernelGGL(lj_scan_kernel<false>, dim3(n_streams), dim3(LJ_T), 0, s, a);
  mark(p, "lj_scan_kernel");
  if (int st = launch_tail(p, a, s, p->any_pipeline))
    return st;
  p->slow_pass_launched = false;
  if (p->any_fast && p->expect_slow) {
    // the last run needed it: no host round trip to find out again
    if (int st = launch_slow_pass(p, s))
      return st;
  }
  if (!p->nk_split.empty())
    return run_nikon_split(p, in_dev, out_dev, s);
  return RSX_OK;
}

int ljpeg_plan_run_(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s,
                    KernelTimer* timer, bool continue_timer) {
  p->last_in = in_dev;
  p->last_out = out_dev;
  struct TimerScope { // the timer covers this run's own launches only
    LJpegPlan* p;
    ~TimerScope() { p->timer = nullptr; }
  } scope{p};
  p->timer = timer;
  if (timer && !continue_timer)
    timer->begin(s);
  if (!p->dri.empty())
    if (int st = run_dri(p, in_dev, out_dev, s))
      return st;
  if (p->streams.empty())
    return RSX_OK;
  if (int st = ljpeg_plan_run_begin_(p, in_dev, out_dev, s))
    return st;
  if (int st = ljpeg_plan_run_blocks_(p, s, 0, p->total_blocks))
    return st;
  return ljpeg_plan_run_end_(p, in_dev, out_dev, s);
}

// ---- a run in chunks (one stream, the single-pass kernel's) ------------------------------
// whether the plan is one the host may run in chunks: ONE stream that the single-pass kernel
// takes, nothing else
bool ljpeg_plan_chunkable(const LJpegPlan* p) {
  return p->streams.size() == 1 && p->streams[0].fast != 0 && p->dri.empty() &&
         p->nk_split.empty() && !p->any_pipeline && !p->any_legacy && !p->any_fast_diffs &&
         !p->dev_layout &&
         !p->expect_slow && p->streams[0].kind == 0u;
  // (kind 0 only: what a prefix of a CR2 stream completes is rows of a vertical STRIP, narrow 2-D
  // copies that run at half the rate of whole rows -- measured on a 6720 x 4480 frame in three strips:
  // 1.94 ms in chunks against 1.87 ms the plain way, profiles/r06/ab/chunked_host_calls.txt)
}
uint32_t ljpeg_plan_blocks(const LJpegPlan* p) { return p->total_blocks; }
int ljpeg_plan_run_begin(LJpegPlan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  p->last_in = in_dev;
  p->last_out = out_dev;
  p->timer = nullptr;
  return ljpeg_plan_run_begin_(p, in_dev, out_dev, s);
}
int ljpeg_plan_run_blocks(LJpegPlan* p, hipStream_t s, uint32_t blk0, uint32_t blk1) {
  return ljpeg_plan_run_blocks_(p, s, blk0, blk1);
}
int ljpeg_plan_run_end(LJpegPlan* p, hipStream_t s) {
  return ljpeg_plan_run_end_(p, p->last_in, p->last_out, s);
}
// the symbols the stream's blocks [0, blk_end) have delivered (waits for the stream)
int ljpeg_plan_symbols_done(LJpegPlan* p, hipStream_t s, uint32_t blk_end, uint64_t* symbols) {
  rsx_ctx* ctx = p->ctx;
  *symbols = 0;
  if (blk_end == 0)
    return RSX_OK;
  uint32_t w[2] = {0, 0};
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(&w[0], static_cast<uint32_t*>(p->d_block_base0.ptr) + (blk_end - 1), 4,
                                    hipMemcpyDeviceToHost, s));
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(&w[1], static_cast<uint32_t*>(p->d_block_sum.ptr) + (blk_end - 1), 4,
                                    hipMemcpyDeviceToHost, s));
  RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
  *symbols = uint64_t(w[0]) + w[1];
  return RSX_OK;
}
// The pixels the stream's symbols [lo, hi) COMPLETE, as rectangles of the image (bytes, rows):
// whole stream rows (LJPEG: a row's kept part) or whole rows of a CR2 strip; a row that the
// range only touches belongs to the range that finishes it.
void ljpeg_plan_region(const LJpegPlan* p, uint64_t lo, uint64_t hi, std::vector<LjRegion>* out) {
  out->clear();
  const LjStreamDev& S = p->streams[0];
  if (hi > S.needed)
    hi = S.needed;
  if (hi <= lo)
    return;
  if (S.kind == 0) {
    const uint64_t r0 = lo / S.row_samples, r1 = hi == S.needed ? S.rows : hi / S.row_samples;
    const uint32_t keep = std::min(S.keep_samples, S.row_samples);
    if (r1 > r0 && keep)
      out->push_back({size_t(S.out_x) * 2, size_t(keep) * 2, size_t(S.out_y) + size_t(r0), size_t(r1 - r0)});
    return;
  }
  for (uint32_t z = 0; z < S.n_strips; ++z) {
    const Cr2Strip& st = p->h_strips[S.strip_base + z];
    const uint64_t f0 = st.first_sample, f1 = p->h_strips[S.strip_base + z + 1].first_sample;
    auto rows_at = [&](uint64_t n) -> uint64_t {
      if (n <= f0)
        return 0;
      if (n >= f1)
        return st.h;
      return std::min<uint64_t>(st.h, (n - f0) / st.w);
    };
    const uint64_t a = rows_at(lo), b = rows_at(hi);
    if (b > a)
      out->push_back({size_t(st.x0) * 2, size_t(st.w) * 2, size_t(st.y0) + size_t(a), size_t(b - a)});
  }
}
// whether the run's streams all came out of the single-pass kernel (call after the results):
// if not, pixels fetched before the end of the run may have been rewritten since
bool ljpeg_plan_single_pass_held(const LJpegPlan* p) {
  for (const LjResult& R : p->h_results)
    if (R.flags & (FL_SLOW | FL_NEED_LEGACY | FL_UNCONVERGED))
      return false;
  return !p->slow_pass_launched;
}

namespace {

// Wait for the last run, fetch the per-stream results and, if a chain was still
// inconsistent after the fixed number of stitch rounds (rare: a start state
// that needs more than one subsequence to synchronise across several
// workgroups), iterate the fix-up to its fixed point and redo the tail.
// Jacobi iteration: at most n_blocks rounds, the fixed point is the serial decode.
int converge(LJpegPlan* p, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  auto fetch = [&]() -> int {
    RSX_HIP_CHECK(ctx, hipMemcpyAsync(p->h_results.data(), results_of_run(p),
                                      p->h_results.size() * sizeof(LjResult),
                                      hipMemcpyDeviceToHost, s));
    if (p->any_fast)
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(p->h_level, p->d_fast_level.ptr, sizeof p->h_level,
                                        hipMemcpyDeviceToHost, s));
    RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
    return RSX_OK;
  };
  if (int st = fetch())
    return st;
  // the LDS level this data needed: the next runs launch that one only
  if (p->any_fast)
    p->level_mask = 1u << std::min(p->h_level[2u + (p->run_count & 1u)], 2u);
  // a child plan whose layout lives on the device (restart intervals): anything the
  // single-pass kernel did not finish is the host-built plan's business (run_dri_host)
  if (p->dev_layout) {
    for (const LjResult& R : p->h_results)
      if (R.flags & (FL_SLOW | FL_UNCONVERGED | FL_NEED_LEGACY))
        return RSX_INTERNAL_RETRY;
    return RSX_OK;
  }
  // streams the single-pass kernel gave up on: the second pass (unless the run
  // launched it already, because the run before needed it)
  if (p->any_fast) {
    bool slow = false;
    for (size_t k = 0; k < p->streams.size(); ++k)
      slow |= p->streams[k].fast && (p->h_results[k].flags & FL_SLOW) != 0;
    p->expect_slow = slow;
    if (slow && !p->slow_pass_launched) {
      if (int st = launch_slow_pass(p, s))
        return st;
      if (int st = fetch())
        return st;
    }
    // A stream the single-pass kernel gave up on twice in a row (the same data decoded
    // again: a batch loop, a benchmark) goes straight to the multi-kernel pipeline from
    // now on; one that it finished is given another chance.
    bool changed = false;
    p->slow_strikes.resize(p->streams.size(), 0);
    for (size_t k = 0; k < p->streams.size(); ++k) {
      if (!p->streams[k].fast)
        continue;
      if (p->h_results[k].flags & FL_SLOW) {
        if (++p->slow_strikes[k] >= 2) {
          p->demoted_fast.resize(p->streams.size(), 0);
          p->demoted_at.resize(p->streams.size(), 0);
          p->demoted_fast[k] = uint8_t(p->streams[k].fast);
          p->demoted_at[k] = p->run_count ? p->run_count : 1u;
          p->streams[k].fast = 0;
          changed = true;
        }
      } else {
        p->slow_strikes[k] = 0;
      }
    }
    if (changed) {
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(p->d_streams.ptr, p->streams.data(),
                                        p->streams.size() * sizeof(LjStreamDev),
                                        hipMemcpyHostToDevice, s));
      RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
      p->any_pipeline = true;
      p->any_fast = false;
      p->any_fast_legacy = false;
      for (const LjStreamDev& S : p->streams) {
        p->any_fast |= S.fast != 0;
        p->any_legacy |= !S.fast && !S.direct;
        p->any_fast_legacy |= S.fast && !S.direct;
      }
      p->expect_slow = false;
      for (size_t k = 0; k < p->streams.size(); ++k)
        p->expect_slow |= p->streams[k].fast && (p->h_results[k].flags & FL_SLOW);
    }
  }
  auto unconverged = [&]() {
    for (const LjResult& R : p->h_results)
      if (R.flags & FL_UNCONVERGED)
        return true;
    return false;
  };
  const uint32_t n_streams = uint32_t(p->streams.size());
  // Fused-path streams whose symbols run past the end of their data (damaged or
  // truncated input): the legacy route knows the reference's end-of-stream
  // semantics.  Its difference scratch for these streams is set up on first use.
  auto legacy_fallback = [&]() -> int {
    bool need = false;
    for (const LjResult& R : p->h_results)
      need |= (R.flags & FL_NEED_LEGACY) != 0;
    if (!need)
      return RSX_OK;
    if (!p->legacy_fallback_ready) {
      for (LjStreamDev& S : p->streams)
        if (S.direct) {
          S.diff_offset = p->total_diffs;
          p->total_diffs += (S.needed + 7 + 8) & ~uint64_t(7);
        }
      if (int st = p->d_diffs.ensure(size_t(p->total_diffs) * 2 + 64))
        return st;
      RSX_HIP_CHECK(ctx, hipMemcpy(p->d_streams.ptr, p->streams.data(),
                                   p->streams.size() * sizeof(LjStreamDev),
                                   hipMemcpyHostToDevice));
      p->legacy_fallback_ready = true;
    }
    LjArgs a2 = make_args(p, p->last_in, p->last_out);
    a2.pass = 2;
    launch_legacy(p, p->fallback, a2, s);
    hipLaunchKernelGGL(lj_consumed_kernel, dim3(n_streams), dim3(64), 0, s, a2);
    RSX_HIP_CHECK(ctx, hipGetLastError());
    return fetch();
  };
  if (!unconverged())
    return legacy_fallback();
  // transfer functions + chain: every workgroup learns its true entry state at
  // once, however badly the stream synchronises; one stitch pass then settles it
  if (int st = p->d_transfer.ensure(size_t(p->total_blocks) * TF_ENTRIES * 2))
    return st;
  LjArgs a = make_args(p, p->last_in, p->last_out);
  a.pass = 2;
  {
    if (p->sync_present[0][0] || p->sync_present[0][1] || p->sync_present[0][2] ||
        p->sync_present[0][4] || p->any_lut11)
      hipLaunchKernelGGL((lj_transfer_kernel<false>), dim3(p->total_blocks), dim3(64),
                         sizeof(TabLds), s, a);
    if (p->any_pair)
      hipLaunchKernelGGL((lj_transfer_kernel<false, true>), dim3(p->total_blocks), dim3(64),
                         sizeof(TabLds), s, a);
    if (p->any_multi)
      hipLaunchKernelGGL((lj_transfer_kernel<true>), dim3(p->total_blocks), dim3(LJ_T),
                         size_t(p->max_tables) * sizeof(TabLds), s, a);
    hipLaunchKernelGGL(lj_chain_kernel, dim3(n_streams), dim3(64), 0, s, a);
    launch_sync<true>(p, a, s);
    hipLaunchKernelGGL(lj_scan_kernel<false>, dim3(n_streams), dim3(LJ_T), 0, s, a);
    RSX_HIP_CHECK(ctx, hipGetLastError());
    p->extra_stitch_rounds += 1;
    if (int st = fetch())
      return st;
  }
  uint32_t rounds = 0;
  // (a stitch launch settles at least LJ_STITCH_MAX_ROUNDS slots of a workgroup that
  // is still chaining, and at least one more workgroup of a chain of workgroups)
#ifndef RSX_STITCH_BOUND_X
#define RSX_STITCH_BOUND_X 1
#endif
  while (unconverged() && rounds <= RSX_STITCH_BOUND_X * (9 * p->total_blocks + 16)) {
    for (int k = 0; k < 4; ++k)
      launch_sync<true>(p, a, s);
    rounds += 4;
    hipLaunchKernelGGL(lj_scan_kernel<false>, dim3(n_streams), dim3(LJ_T), 0, s, a);
    RSX_HIP_CHECK(ctx, hipGetLastError());
    if (int st = fetch())
      return st;
  }
  p->extra_stitch_rounds += int(rounds);
  // the optimistic tail ran on an inconsistent chain: forget what it reported
  // (not for the streams the single-pass kernel finished: nothing is redone for them)
  for (size_t k = 0; k < p->h_results.size(); ++k) {
    LjResult& R = p->h_results[k];
    if (p->streams[k].fast && !(R.flags & FL_SLOW))
      continue;
    R.status = 0;
    R.tail_used = 0;
    R.last_slot = R.last_pos = R.consumed = 0;
  }
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(results_of_run(p), p->h_results.data(),
                                    p->h_results.size() * sizeof(LjResult),
                                    hipMemcpyHostToDevice, s));
  if (int st = launch_tail(p, a, s, true, 3))
    return st;
  if (int st = fetch())
    return st;
  return legacy_fallback();
}

} // namespace

int ljpeg_plan_results(LJpegPlan* p, hipStream_t s, bool ran, int32_t* job_status,
                       uint32_t* job_consumed) {
  int rc = RSX_OK;
  if (ran && !p->streams.empty())
    if (int st = converge(p, s))
      return st;
  // NikonDecompressor jobs with a split: the rows after it are the child's jobs
  std::vector<int32_t> nk_status(p->nk_split.size(), RSX_OK);
  if (ran && !p->nk_split.empty()) {
    std::vector<int32_t> cst;
    if (p->nk_child) {
      cst.assign(p->nk_child->n_jobs, RSX_OK);
      const int crc = ljpeg_plan_results(p->nk_child, s, true, cst.data(), nullptr);
      if (crc == RSX_ERR_DEVICE || crc == RSX_ERR_NOMEM)
        return crc;
    }
    for (size_t k = 0; k < p->nk_split.size(); ++k)
      nk_status[k] = p->nk_split[k].status;
    for (size_t c = 0; c < p->nk_child_owner.size(); ++c)
      if (nk_status[p->nk_child_owner[c]] == RSX_OK)
        nk_status[p->nk_child_owner[c]] = cst[c];
  }
  // restart-interval jobs: fold the child plan's per-interval results
  std::vector<int32_t> dri_status(p->dri.size(), RSX_OK);
  std::vector<uint32_t> dri_consumed(p->dri.size(), 0);
  if (ran && !p->dri.empty()) {
    std::vector<int32_t> cst;
    std::vector<uint32_t> ccons;
    rsx_ctx* ctx = p->ctx;
    if (p->dri_ran_dev) {
      // The device-laid-out child plan: markers, statuses and the child's results arrive
      // together, AFTER everything was launched.  Anything irregular -- markers missing, a
      // list too short for the FFxx of the scan, a stream the single-pass kernel gave up
      // on -- and the job is redone the old way, by the host-built plan.
      const size_t nd = p->dri.size();
      std::vector<uint32_t> counts(nd), status(nd + 1);
      size_t total_sorted = 0;
      for (const DriJobDev& D : p->dri_dev)
        total_sorted += D.n_ri;
      std::vector<uint2> sorted(total_sorted);
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(counts.data(), p->d_marker_count.ptr, nd * 4,
                                        hipMemcpyDeviceToHost, s));
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(status.data(), p->d_dri_status.ptr, (nd + 1) * 4,
                                        hipMemcpyDeviceToHost, s));
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(sorted.data(), p->d_dri_sorted.ptr,
                                        total_sorted * sizeof(uint2), hipMemcpyDeviceToHost, s));
      RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
      bool ok = status[nd] == 0;
      for (size_t d = 0; d < nd && ok; ++d)
        ok = status[d] == 0 && counts[d] <= p->dri_dev[d].cap && counts[d] + 1 >= p->dri[d].n_ri;
      if (ok) {
        LJpegPlan* c = p->child_dev;
        for (size_t d = 0; d < nd; ++d) {
          auto& dj = p->dri[d];
          const DriJobDev& D = p->dri_dev[d];
          dj.status = RSX_OK;
          dj.markers.clear();
          dj.codes.clear();
          dj.starts.assign(1, 0u);
          for (uint32_t i = 0; i + 1 < dj.n_ri; ++i) {
            const uint2 m = sorted[D.sorted_off + i];
            dj.markers.push_back(m.x);
            dj.codes.push_back(uint8_t(m.y));
            dj.starts.push_back(m.x + 2);
          }
          // (the host's copy of the child's stream records, as the layout kernel wrote them:
          // the checks below compare consumed bytes with the streams' sizes)
          for (uint32_t i = 0; i < dj.n_ri; ++i) {
            LjStreamDev& S = c->streams[D.first_stream + i];
            const uint64_t start = dj.starts[i];
            uint64_t end = (i + 1 < dj.n_ri) ? uint64_t(dj.markers[i]) + 2 : D.in_bytes;
            end = std::min<uint64_t>(D.in_bytes, std::max<uint64_t>(end, start + 8));
            S.in_offset = D.in_offset + start;
            S.in_bytes = end - std::min(start, end);
          }
        }
        cst.assign(c->n_jobs, RSX_OK);
        ccons.assign(c->n_jobs, 0);
        const int crc = ljpeg_plan_results(c, s, true, cst.data(), ccons.data());
        if (crc == RSX_ERR_DEVICE || crc == RSX_ERR_NOMEM)
          return crc;
        if (crc == RSX_INTERNAL_RETRY)
          ok = false;
      }
      if (!ok) {
        p->dri_ran_dev = false;
        if (int st = run_dri_host(p, p->last_in, p->last_out, s))
          return st;
      }
    }
    if (!p->dri_ran_dev && p->child) {
      cst.assign(p->child->n_jobs, RSX_OK);
      ccons.assign(p->child->n_jobs, 0);
      const int crc = ljpeg_plan_results(p->child, s, true, cst.data(), ccons.data());
      if (crc == RSX_ERR_DEVICE || crc == RSX_ERR_NOMEM)
        return crc;
    }
    for (size_t d = 0; d < p->dri.size(); ++d)
      dri_status[d] = p->dri[d].status;
    const auto& owner = p->dri_ran_dev ? p->child_dev_owner : p->child_owner;
    for (size_t c = 0; c < owner.size(); ++c) {
      const int d = owner[c].first;
      const uint32_t i = uint32_t(owner[c].second);
      const auto& dj = p->dri[d];
      if (dri_status[d] != RSX_OK)
        continue; // the first failing interval decides (the reference stops there)
      if (i > 0) {
        // marker in front of interval i must be RST((i-1) % 8) (:288-297)
        const uint8_t code = dj.codes[i - 1];
        if (code < 0xD0 || code > 0xD7 || uint32_t(code - 0xD0) != ((i - 1) % 8)) {
          dri_status[d] = RSX_ERR_RESTART_MARKER;
          continue;
        }
      }
      if (cst[c] != RSX_OK) {
        dri_status[d] = cst[c];
        continue;
      }
      if (i + 1 < dj.n_ri && dj.starts[i] + ccons[c] != dj.markers[i]) {
        // the interval must end exactly on its marker (:289-291, :335)
        dri_status[d] = RSX_ERR_RESTART_MARKER;
        continue;
      }
      dri_consumed[d] = dj.starts[i] + ccons[c];
    }
    for (size_t d = 0; d < p->dri.size(); ++d)
      if (dri_status[d] == RSX_OK && dri_consumed[d] > p->dri[d].in.geom.in_bytes)
        dri_status[d] = RSX_ERR_IO;
  }
#ifdef RSX_EXPERIMENT
  if (getenv("RSX_DEBUG") && p->d_dbg.ptr && ran) {
    // mean duration of the single-pass kernel's phases (shader clock ticks -> us at 2.4 GHz)
    std::vector<unsigned long long> t(size_t(p->total_blocks) * 16);
    if (hipMemcpy(t.data(), p->d_dbg.ptr, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      double sum[16] = {}, mx[16] = {};
      std::vector<float> all[16];
      size_t n = 0;
      unsigned long long tmin = ~0ull, tmax = 0;
      for (uint32_t b = 0; b < p->total_blocks; ++b) {
        const unsigned long long* r = &t[size_t(b) * 16];
        if (!r[0] || !r[15])
          continue;
        ++n;
        tmin = std::min(tmin, r[0]);
        tmax = std::max(tmax, r[15]);
        unsigned long long prev = r[0];
        for (int k = 1; k < 16; ++k) {
          const unsigned long long cur = r[k] ? r[k] : prev;
          const double d = double(cur - prev) / 2400.0;
          sum[k] += d;
          mx[k] = std::max(mx[k], d);
          all[k].push_back(float(d));
          prev = cur;
        }
      }
      static const char* nm[16] = {"", "ticket+stream", "tables+image", "bit delay", "guess", "decode",
                                   "rounds+scan", "symbol base", "fetch+records", "geometry+barrier",
                                   "staging", "rows + scan", "lb1", "C table",
                                   "-", "copy-out"};
      fprintf(stderr, "[rsx] single-pass phases over %zu workgroups, kernel span %.1f us:\n", n,
              double(tmax - tmin) / 2400.0);
      double tot = 0;
      for (int k = 1; k < 16; ++k) {
        std::sort(all[k].begin(), all[k].end());
        auto pct = [&](double q) { return all[k].empty() ? 0.0 : double(all[k][size_t(q * (all[k].size() - 1))]); };
        fprintf(stderr, "[rsx]   %-18s mean %7.2f us  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %8.2f us\n",
                nm[k], n ? sum[k] / n : 0.0, pct(0.5), pct(0.9), pct(0.99), mx[k]);
        tot += n ? sum[k] / n : 0.0;
      }
      fprintf(stderr, "[rsx]   %-14s mean %7.2f us\n", "lifetime", tot);
    }
    {
      // ... and of lj_unstuff_kernel's (second half of the array)
      std::vector<unsigned long long> t0(size_t(p->total_blocks) * 16);
      if (hipMemcpy(t0.data(), static_cast<unsigned long long*>(p->d_dbg.ptr) + size_t(p->total_blocks) * 16,
                    t0.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        static const char* nm0[10] = {"", "loads+park", "list", "un-stuff", "write-out+barrier",
                                      "tables+A parse", "B parse", "rounds", "hand-over+count", "word+level"};
        std::vector<float> all[10];
        unsigned long long tmin = ~0ull, tmax = 0;
        size_t n = 0;
        for (uint32_t b = 0; b < p->total_blocks; ++b) {
          const unsigned long long* r = &t0[size_t(b) * 16];
          if (!r[0] || !r[9])
            continue;
          ++n;
          tmin = std::min(tmin, r[0]);
          tmax = std::max(tmax, r[9]);
          unsigned long long prev = r[0];
          for (int k = 1; k < 10; ++k) {
            const unsigned long long cur = r[k] ? r[k] : prev;
            all[k].push_back(float(double(cur - prev) / 2400.0));
            prev = cur;
          }
        }
        fprintf(stderr, "[rsx] K0 phases over %zu workgroups, kernel span %.1f us:\n", n,
                n ? double(tmax - tmin) / 2400.0 : 0.0);
        double tot = 0;
        for (int k = 1; k < 10 && n; ++k) {
          std::sort(all[k].begin(), all[k].end());
          double sum = 0;
          for (float v : all[k])
            sum += v;
          auto pct = [&](double q) { return double(all[k][size_t(q * (all[k].size() - 1))]); };
          fprintf(stderr, "[rsx]   K0 %-18s mean %7.2f us  p50 %6.2f  p90 %6.2f  p99 %6.2f  max %8.2f us\n",
                  nm0[k], sum / n, pct(0.5), pct(0.9), pct(0.99), double(all[k].back()));
          tot += sum / n;
        }
        fprintf(stderr, "[rsx]   K0 %-14s mean %7.2f us\n", "lifetime", tot);
      }
    }
    // K0's words: how many workgroups does the single-pass kernel have to ask for their count?
    std::vector<unsigned long long> kw(size_t(p->total_blocks) + 1);
    if (p->d_k0w.ptr &&
        hipMemcpy(kw.data(), p->d_k0w.ptr, kw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      size_t nz = 0, unc = 0, differ = 0, norec = 0;
      for (uint32_t b = 0; b < p->total_blocks; ++b) {
        const unsigned long long w = kw[b];
        if (!w)
          continue;
        ++nz;
        const uint32_t own = uint32_t(w >> 32) & 0xFFFFu, tru = uint32_t(w >> 48);
        unc += (own >> 8) & 1u;
        norec += !(tru & 0x8000u);
        differ += (tru & 0x8000u) && (own & 0xFFu) != (tru & 0xFFu);
      }
      fprintf(stderr,
              "[rsx] K0 words: %zu workgroups, %zu uncertain, %zu entry estimate != true entry, "
              "%zu without a true entry\n",
              nz, unc, differ, norec);
    }
  }
  if (getenv("RSX_DEBUG")) {
    fprintf(stderr, "[rsx] ljpeg plan: %zu streams, extra stitch rounds %d\n",
            p->streams.size(), p->extra_stitch_rounds);
    for (size_t k = 0; k < p->h_results.size(); ++k) {
      const LjResult& R = p->h_results[k];
      fprintf(stderr,
              "[rsx]  stream %zu: marker %u status %u flags %u avail %u needed %llu "
              "last_slot %u last_pos %u consumed %u tail %u blocks %u in_bytes %llu "
              "redo_rounds %u redo_slots %u not_merged+stitched %u max_rounds %u (workgroup %u%s) "
              "single-pass gave up: reasons 0x%x (block %u slot %u symbols before it %u base %u; scan: 0x%x %u %u)\n",
              k, R.marker_pos, R.status, R.flags, R.avail_lo,
              (unsigned long long)p->streams[k].needed, R.last_slot, R.last_pos,
              R.consumed, R.tail_used, p->streams[k].n_blocks,
              (unsigned long long)p->streams[k].in_bytes, R.stat_rounds, R.stat_redo,
              R.stat_stitch, R.pad2 >> 16, R.pad2 & 0x7FFFu, (R.pad2 & 0x8000u) ? ", stitch" : "",
              R.stat_why, R.pad3[0], R.pad3[1] & 0xFFFFu, R.pad3[1] >> 16, R.pad3[2], R.pad3[0],
              R.pad3[1], R.pad3[2]);
    }
  }
#endif
  for (int i = 0; i < p->n_jobs; ++i) {
    int st = p->job_status[i];
    uint32_t consumed = 0;
    for (size_t d = 0; d < p->dri.size(); ++d)
      if (p->dri[d].job == i && st == RSX_OK && ran) {
        st = dri_status[d];
        consumed = dri_consumed[d];
      }
    if (st == RSX_OK && ran && p->job_first_stream[i] >= 0) {
      const int fs = p->job_first_stream[i];
      for (int k = 0; k < p->job_n_streams[i]; ++k) {
        const LjResult& R = p->h_results[fs + k];
        const LjStreamDev& S = p->streams[fs + k];
        if (R.flags & FL_UNCONVERGED)
          st = RSX_ERR_DEVICE; // cannot happen: the loop above runs to the fixed point
        else if (R.status != 0)
          st = int(R.status);
        else if (uint64_t(R.avail_lo) < S.needed)
          st = RSX_ERR_INPUT_OVERFLOW;
        consumed = R.consumed;
        if (st == RSX_OK && S.kind == 0 && !S.pair && uint64_t(consumed) > S.in_bytes)
          st = RSX_ERR_IO; // inputStream.skipBytes(): LJpegDecompressor.cpp:335
        // SonyArw1: the codes missing from the table are the lengths 13..17, whose
        // differences always take the value out of range (SonyArw1Decompressor.cpp:88)
        if (st == RSX_ERR_BAD_HUFFMAN_CODE && S.kind == 2 && p->nk[fs + k].sony)
          st = RSX_ERR_VALUE_RANGE;
      }
    }
    for (size_t k = 0; k < p->nk_split.size(); ++k)
      if (p->nk_split[k].job == i && st == RSX_OK && ran)
        st = nk_status[k];
    if (job_status)
      job_status[i] = st;
    if (job_consumed)
      job_consumed[i] = st == RSX_OK ? consumed : 0;
    if (st != RSX_OK)
      rc = st;
  }
  return rc;
}

void ljpeg_plan_destroy(LJpegPlan* p) {
  if (!p)
    return;
  if (p->child)
    ljpeg_plan_destroy(p->child);
  if (p->child_dev)
    ljpeg_plan_destroy(p->child_dev);
  if (p->nk_child)
    ljpeg_plan_destroy(p->nk_child);
  for (DeviceBuffer* b : {&p->d_nk, &p->d_nk_tables, &p->d_nk_rowpow, &p->d_nk_pup,
                          &p->d_transfer, &p->d_fast_tabs, &p->d_lb, &p->d_tickets, &p->d_dbg, &p->d_fast_z, &p->d_fast_level,
                          &p->d_k0w, &p->d_k0e, &p->d_k0p, &p->d_block_base0,
                          &p->d_fast_order})
    b->release();
  p->d_marker_count.release();
  p->d_marker_list.release();
  for (DeviceBuffer* b :
       {&p->d_streams, &p->d_tables, &p->d_block_stream, &p->d_strips,
        &p->d_block_flags, &p->d_block_tf, &p->d_sub_start, &p->d_sub_state, &p->d_sub_sums, &p->d_sub_first, &p->d_sub_psum, &p->d_block_start, &p->d_block_exit,
        &p->d_block_sum, &p->d_block_base, &p->d_block_psum, &p->d_block_pbase,
        &p->d_block_drops, &p->d_block_drop_base, &p->d_results, &p->d_diffs, &p->d_vseed,
        &p->d_row_edge, &p->d_unstuffed})
    b->release();
  delete p;
}

} // namespace rsx
