// rsx_ljpeg_fast.hip -- single-pass lossless-JPEG decode for gfx950: ONE kernel turns
// the un-stuffed entropy stream of a clean single-table LJPEG / CR2 scan into pixels.
//
// Replaces, for these streams, the walk of
//   LJpegDecompressor::decodeN / decodeRowN  (decompressors/LJpegDecompressor.cpp:184-251, 300-332)
//   Cr2Decompressor::decompressN_X_Y         (decompressors/Cr2DecompressorImpl.h:419-465)
// and the multi-kernel pipeline of rsx_ljpeg.hip / rsx_ljpeg_direct.hip (synchronisation
// with a recorded pass, two stitch passes, scan, row edges, row offsets, final decode:
// every symbol parsed three times at ~131 lane-instructions per symbol), which stays
// in the library as the slow path: streams the single pass cannot finish are flagged
// FL_SLOW and redone by it (LjArgs::pass == 1).
//
// Per workgroup (256 lanes = 255 own 64-byte subsequences + a copy of the predecessor's
// last one; workgroups take TICKETS -- the streams' blocks interleaved -- so that every
// predecessor is resident or finished):
//  1. start guesses: made by lj_unstuff_kernel (rsx_ljpeg.hip: a parse of the three
//     subsequences before each one from bit 0 -- Huffman streams self-synchronise --, the
//     symbol grid of constant runs from their bits, and the LDS level of the launch:
//     the kernel is launched at up to three allocations and the one whose level the
//     data needs does the work).  Image, table, guesses and the stream's flags are asked
//     for as soon as the ticket names the block.
//  2. decode: every lane decodes its subsequence ONCE from its guess and KEEPS the
//     running sums of its differences (by component phase, packed 2 x 16 bit) in 64
//     VGPRs: 16 VALU instructions per symbol.  A lane that meets a code longer than
//     the 10-bit LUT, an invalid code or SSSS = 16 stops there.
//  3. Jacobi rounds: subsequences whose guess was not their predecessor's exit, or that
//     stopped, are re-decoded by the first lanes into an LDS side buffer (entries handed
//     out in slot order); after the first round only the head of a run of inconsistent
//     slots is redone.  What the rounds cannot finish only matters if a delivered symbol
//     lies behind it (trailing bytes, padding rows: nobody's business).
//  4. look-back 0 (decoupled, one 8-byte granule per workgroup: assumed entry state,
//     exit state, symbols, inclusive symbol base; published as soon as the three are
//     known): the workgroup's entry state is checked against its predecessor's exit (a
//     wrong one is repaired and published again) and its first symbol's index comes out
//     of the walk.
//  5. staging: ALL running sums of the workgroup go to LDS in stream order, over the LUT,
//     the image and the records, which are dead by then.
//  6. rows: the lanes read, for the stream rows that start in the workgroup, the first
//     MCU's (running sum before it, its difference) from the staged samples; a scan over
//     the rows gives the workgroup's transfer of the predictor state
//        (T = left-neighbour values, Vc = first-MCU values of the last started row),
//     look-back 1 carries that state across workgroups.
//  7. output: the whole workgroup writes the staged samples out, run by run (row, kept
//     width, CR2 strip), as 16-byte stores on the destination's 16-byte grid:
//     pixel = staged sample + constant of (row, component).
// A stream the kernel cannot finish (periodic data that is not a run of the zero code,
// invalid codes, fewer than ~4 bits per symbol) is flagged FL_SLOW.
//
// Arithmetic (everything mod 2^16; cf. tests/test_direct_recon_model.py): with Ploc(i)
// the running sum of i's component over the workgroup's symbols up to i,
//   X(i) = Ploc(i) + C(row(i), comp(i)),
//   C(r, c) = Vc_in[c] + sum of the first-MCU differences D(r', c) of the rows r' < r
//             that start inside the workgroup - Ploc before (r, c)'s first-MCU symbol,
//   C(r, c) = T_in[c] for the row that is open when the workgroup starts.
//
// Instruction selection follows scripts/ubench/valu_rates2.hip (profiles/r03/): on
// gfx950 add / sub / and / or / xor / lshr / ashr / mov issue in ~2.4 cycles per
// wave64, everything else (lshl, bfe, alignbit, mad, SDWA, packed, v_cmp) in ~4.3.
// The position is kept as Pn = -32 * pos - 32 so that the window's LDS row is an AND +
// ADD and its shift amount a logical shift right; the un-stuffed image lies in LDS
// delayed by one bit so that v_alignbit's 5-bit amount (31 - pos % 32) selects the
// window exactly.  No MFMA: there is no contraction anywhere.
#include "rsx_ljpeg_bits.h"

namespace rsx {

namespace {

// (experiment, round 6: -DRSX_LF_WG_PER_CU=5 makes the compiler fit the kernel into the 96 vector
// registers five workgroups a CU would leave it -- the LDS still holds four: what the registers
// alone would cost, profiles/r06/ab/five_workgroups_register_cap.txt)
#ifdef RSX_LF_WG_PER_CU
constexpr int LF_WG_PER_CU = RSX_LF_WG_PER_CU;
#else
constexpr int LF_WG_PER_CU = 4;
#endif
constexpr int LF_BW = LJ_PW + 1;        // dword rows of a subsequence the loops can touch
constexpr int LF_MAXSYM = 128;          // symbols a lane keeps (64 VGPRs)
constexpr int LF_NR = LF_MAXSYM / 2;
constexpr int LF_NSIDE = 24;            // side-buffer entries (re-decodes per workgroup): what the
                                        // 40 KB of four workgroups a CU leave (16 until round 4: a CR2
                                        // whose table gives large differences long codes stops a lane at
                                        // every strip-row jump, ~12 a workgroup)
constexpr int LF_SIDE_STRIDE = 272;     // bytes: 128 x u16 + 16 (16-byte aligned rows)
constexpr int LF_RMAX = 256;            // stream rows that may start inside one workgroup
constexpr uint32_t LF_MAX_ROUNDS = 6;   // re-decode rounds before the stream is given up
constexpr int LF_K0_IT = 16;            // K0 words a lane asks for at once (4096 workgroups)
constexpr uint32_t LF_SPIN_LIMIT = 1u << 17;    // passes of look-back 1 (~2 us each: a quarter of a second)
constexpr uint32_t LF_SPIN_LIMIT_K0 = 1u << 15; // polls of a flagged predecessor's granule (~20 ms)
// Ablation switches of experiment builds (scripts/exp_ab.py; wrong pixels, timing only):
// 1 no staging + copy-out, 2 no copy-out, 4 no look-back 0, 8 no look-back 1, 16 no
// decode loop, 32 no warm-up, 64 no re-decode rounds, 128 no row table
#if defined(RSX_EXPERIMENT) && defined(RSX_LF_ABLATE)
constexpr uint32_t LF_ABLATE = RSX_LF_ABLATE;
#else
constexpr uint32_t LF_ABLATE = 0;
#endif

// ---- LDS layout (bytes) ------------------------------------------------------
// (The LUT is addressed ABSOLUTELY -- (window >> 19) & 0x1FF8 is the LDS address of its
// entry --, so the kernel's dynamic LDS has to start at LDS address 0: nothing in it may
// bring static LDS along.  __syncthreads_or() does -- 256 bytes, .amdhsa_group_segment_
// fixed_size -- and with it every look-up read 256 bytes off: wrong symbols, broken
// look-back links, workgroups spinning to their limits.)
constexpr uint32_t LF_OFF_LUT = 0;                       // 1024 x uint2: LDS address 0
constexpr uint32_t LF_OFF_B = 8192;
constexpr uint32_t LF_OFF_REC = LF_OFF_B + LF_BW * LJ_T * 4;   // u32[256]
constexpr uint32_t LF_OFF_OB = LF_OFF_REC + LJ_T * 4;          // u16[256]
constexpr uint32_t LF_OFF_SM = LF_OFF_OB + LJ_T * 2;           // uint2[256]
constexpr uint32_t LF_OFF_LIST = LF_OFF_SM + LJ_T * 8;         // u16[256]
constexpr uint32_t LF_OFF_SIDE = LF_OFF_LIST + LJ_T * 2;
constexpr uint32_t LF_OFF_FIXED_END = LF_OFF_SIDE + LF_NSIDE * LF_SIDE_STRIDE;
// ... and at the END of the allocation (its size is a launch parameter, LjArgs::fast_lds):
// misc[128], the C table (uint2[LF_RMAX]), the stream's CR2 strips
constexpr uint32_t LF_TAIL_MISC = 0, LF_TAIL_CTAB = 512, LF_TAIL_STRIPS = 512 + LF_RMAX * 8;
constexpr uint32_t LF_TAIL_BYTES =
    (LF_TAIL_STRIPS + (MAX_CR2_STRIPS + 1) * uint32_t(sizeof(Cr2Strip)) + 15u) & ~15u;
// Output staging: when the pixels are staged, everything in front of the tail is dead --
// LUT, image, records, side buffer -- and ALL of the workgroup's running sums are staged
// there at once, in stream order (34 KB = 17 000 samples with the smallest allocation;
// streams with more symbols per byte get a larger one, or the slow path).  Then the
// registers that held them are free for the rest of the kernel, and the first-MCU symbols
// of the stream rows are read from the staged samples.  (Earlier versions kept the 64
// registers through both look-backs and staged wavefront by wavefront: spills in the
// decode loop at 128 VGPRs, and 4 x (stage, barrier, copy, barrier).)
constexpr uint32_t LF_STAGE_BASE = 16; // (a partial first chunk reads up to 14 bytes before)
constexpr uint32_t LF_LDS_MIN = LF_OFF_FIXED_END + LF_TAIL_BYTES;
static_assert(LF_OFF_B % 16 == 0 && LF_OFF_SIDE % 16 == 0 && LF_OFF_FIXED_END % 16 == 0,
              "16-byte aligned regions");
static_assert(4 * ((LF_LDS_MIN + 1279) / 1280) * 1280 <= 160 * 1024, "four workgroups per CU");

// misc[] indices
enum : int {
  M_WCNT = 0,   // [4] symbols per wavefront
  M_WSUM = 4,   // [8] difference sums per wavefront (uint2 x 4)
  M_LIST = 12,  // re-decode list length
  M_TICKET = 13,
  M_UNRESB = 14, // symbols in front of slot misc[M_UNRES]
  M_PRED = 15,  // predecessor's exit state
  M_SLOW = 16,  // != 0: give the stream to the slow path
  M_UNRES = 17, // first slot the kernel could not finish (0xFFFF: none)
  M_WNE = 18,   // [4] side entries the wavefronts ask for in a round
  M_RSUM = 22,  // [8] row scan: per-wavefront totals (uint2 x 4)
  M_LB1 = 30,   // [8] the LOCAL record (a, v) / (T_out, Vc_out)
  M_NSIDE = 38, // side-buffer entries handed out
  M_LBX = 40,   // [48] look-back: per-wavefront window summaries
};

struct FastLds {
  uint8_t* base;
  uint32_t* B;
  uint32_t* rec;
  uint16_t* ob;
  uint2* sm;
  uint16_t* list;
  uint32_t* misc;
  uint8_t* side;
  uint2* ctab;    // C(r, c) of the stream rows the workgroup touches
  uint8_t* strips;
  uint32_t stage_cap; // samples the staging region holds
};

__device__ __forceinline__ FastLds carve_fast(uint8_t* smem, uint32_t lds_bytes) {
  FastLds f;
  uint8_t* tail = smem + (lds_bytes - LF_TAIL_BYTES);
  f.base = smem;
  f.B = reinterpret_cast<uint32_t*>(smem + LF_OFF_B);
  f.rec = reinterpret_cast<uint32_t*>(smem + LF_OFF_REC);
  f.ob = reinterpret_cast<uint16_t*>(smem + LF_OFF_OB);
  f.sm = reinterpret_cast<uint2*>(smem + LF_OFF_SM);
  f.list = reinterpret_cast<uint16_t*>(smem + LF_OFF_LIST);
  f.misc = reinterpret_cast<uint32_t*>(tail + LF_TAIL_MISC);
  f.side = smem + LF_OFF_SIDE;
  f.ctab = reinterpret_cast<uint2*>(tail + LF_TAIL_CTAB);
  f.strips = tail + LF_TAIL_STRIPS;
  f.stage_cap = (lds_bytes - LF_TAIL_BYTES - LF_STAGE_BASE - 16u) / 2u; // (ljpeg_fast_stage_cap)
  return f;
}

typedef uint32_t lf_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lf_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) lf_u32x2* lds_u2p;
typedef __attribute__((address_space(3))) uint16_t* lds_u16w;
typedef __attribute__((address_space(3))) uint32_t* lds_u32w;

// a value every lane holds (read from LDS after a barrier): tell the compiler, so that
// what is derived from it lives in SGPRs
__device__ __forceinline__ uint32_t uni(uint32_t x) {
  return uint32_t(__builtin_amdgcn_readfirstlane(int(x)));
}

__device__ __forceinline__ uint32_t pack16(uint32_t lo, uint32_t hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u); // {hi[15:0], lo[15:0]}
}
// Wavefront scans on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes,
// row_bcast 15 / 31 across them; lanes without a source get `old` = 0, the identity of
// every operator used here): six dependent VALU instructions.  The __shfl_up versions
// they replace compile to six dependent ds_bpermute_b32 -- an LDS round trip each, ~0.3 us
// a scan, ~2 us of a workgroup's 30 over its scans.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t x) {
  return uint32_t(__builtin_amdgcn_update_dpp(0, int(x), CTRL, ROW_MASK, 0xF, false));
}
#define LF_DPP_SCAN(STEP)     \
  STEP(0x111, 0xF)            \
  STEP(0x112, 0xF)            \
  STEP(0x114, 0xF)            \
  STEP(0x118, 0xF)            \
  STEP(0x142, 0xA)            \
  STEP(0x143, 0xC)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x, int) {
#define LF_STEP_ADD(C, M) x += dpp0<C, M>(x);
  LF_DPP_SCAN(LF_STEP_ADD)
#undef LF_STEP_ADD
  return x;
}
__device__ __forceinline__ uint2 wave_scan_pk2(uint2 x, int) {
#define LF_STEP_PK(C, M) x = make_uint2(pk_add(x.x, dpp0<C, M>(x.x)), pk_add(x.y, dpp0<C, M>(x.y)));
  LF_DPP_SCAN(LF_STEP_PK)
#undef LF_STEP_PK
  return x;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) { // (uniform result)
#define LF_STEP_MAX(C, M) x = max(x, dpp0<C, M>(x));
  LF_DPP_SCAN(LF_STEP_MAX)
#undef LF_STEP_MAX
  return uint32_t(__builtin_amdgcn_readlane(int(x), 63));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) { // (uniform result)
  return uint32_t(__builtin_amdgcn_readlane(int(wave_scan_u32(x, 0)), 63));
}
// index mod N (N = 1, 2, 4: a mask; N = 3 -- round 5 -- a division by a constant)
template <int N>
__device__ __forceinline__ uint32_t lf_mod(uint32_t x) {
  return (N & (N - 1)) == 0 ? (x & uint32_t(N - 1)) : x % uint32_t(N);
}
// 16-bit field q (0..3) of a packed uint2
__device__ __forceinline__ uint32_t fld(uint2 v, uint32_t q) {
  return (((q & 2u) ? v.y : v.x) >> (16u * (q & 1u))) & 0xFFFFu;
}
// the fields whose flag bit is set, as 16-bit masks
__device__ __forceinline__ uint2 fld_mask(uint32_t flags) {
  return make_uint2(((flags & 1u) ? 0xFFFFu : 0u) | ((flags & 2u) ? 0xFFFF0000u : 0u),
                    ((flags & 4u) ? 0xFFFFu : 0u) | ((flags & 8u) ? 0xFFFF0000u : 0u));
}
__device__ __forceinline__ uint2 sel2(uint2 m, uint2 x, uint2 y) { // m ? x : y, field-wise
  return make_uint2((x.x & m.x) | (y.x & ~m.x), (x.y & m.y) | (y.y & ~m.y));
}

// ---- look-back granules ----------------------------------------------------------
typedef unsigned long long u64;
__device__ __forceinline__ void lb_store(u64* p, u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// words 1..8: bit 63 valid | flags (8 bits at 32) | payload (two 16-bit fields)
constexpr u64 LB_VALID = 1ull << 63;
// where the words of component pair k sit in a record: LOCAL transfer (a, v), inclusive state (t, c)
#ifdef RSX_LF_LB16
// (experiment: a record of 80 bytes, every (a, v) and (t, c) pair on a 16-byte boundary -- one
// 16-byte load a pair; every 8-byte word carries its own valid bit, so a pair whose halves
// were written apart is seen as what it is)
__device__ __forceinline__ constexpr int lb_wa(int k) { return 2 + 4 * k; }
__device__ __forceinline__ constexpr int lb_wv(int k) { return 3 + 4 * k; }
__device__ __forceinline__ constexpr int lb_wt(int k) { return 4 + 4 * k; }
__device__ __forceinline__ constexpr int lb_wc(int k) { return 5 + 4 * k; }
#else
__device__ __forceinline__ constexpr int lb_wa(int k) { return 1 + k; }
__device__ __forceinline__ constexpr int lb_wv(int k) { return 3 + k; }
__device__ __forceinline__ constexpr int lb_wt(int k) { return 5 + k; }
__device__ __forceinline__ constexpr int lb_wc(int k) { return 7 + k; }
#endif
// word 0 since round 4: bit 63 valid | (symbols the workgroup decoded - symbols K0 counted
// for it), 32 bits: what a successor adds to K0's count of a FLAGGED workgroup.
// K0's word of a workgroup (LjArgs::k0w, lj_unstuff_kernel): symbols (32) | own estimate of
// the entry state, bit 7: count uncertain (16) | true entry state, bit 15: on record (16).
__device__ __forceinline__ bool lf_k0_flagged(u64 w) {
  const uint32_t own = uint32_t(w >> 32) & 0xFFFFu, tru = uint32_t(w >> 48);
  return w != 0ull && (own != (tru & 0xFFu) || !(tru & 0x8000u));
}

// ---------------------------------------------------------------------------
// The fast loops.  LUT entry (uint2; built by ljpeg_build_fast_table):
//   x: bits 0..4  shift that right-aligns the symbol's difference bits (32 - total)
//      bits 5..10 total bits of the symbol, i.e. 32 * total at bit 0
//      bit 31     special: code longer than 10 bits, invalid code, SSSS = 16
//   y: 2^SSSS - 1
// ---------------------------------------------------------------------------
struct FastState {
  uint32_t Pn;   // -32 * pos - 32
  uint32_t n;    // symbols decoded
  uint32_t acc[4];
  uint32_t ev;   // running sum after the last even-numbered symbol
  uint32_t spec; // (per-phase tables) OR of the entries read: bit 15 = a special one among them
#ifdef RSX_LF_PAIRWIN
  uint32_t w, wb, sh; // (experiment: the even step's 64-bit window and its entry's shift, for the odd step)
#endif
};

// (DIFF -- round 6, streams whose reconstruction stays with the legacy kernels: the lane keeps
// the symbols' DIFFERENCES, not their running sums: an assignment where the sum is an addition)
template <int N, int K, bool DIFF = false>
__device__ __forceinline__ void lf_step(FastState& s, uint32_t vbase) {
  const uint32_t ad = vbase + (s.Pn & ~1023u);
#ifdef RSX_LF_SPLIT_READS
  // (two ds_read_b32: 56 cycles where the ds_read2st64_b32 the compiler makes of them takes
  // 73, scripts/ubench/valu_rates.hip)
  uint32_t d0, d1;
  asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(d1), "=&v"(d0)
               : "v"(ad));
#else
  const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
#endif
  const uint32_t w = __builtin_amdgcn_alignbit(d0, d1, s.Pn >> 5);
  const lf_u32x2 e = *(lds_u2p)((w >> 19) & 0x1FF8u);
  const uint32_t v = (w >> (e.x & 31u)) & e.y;
  // JPEG EXTEND without a shift left: u = all - v; t = u - v < 0 iff the top
  // difference bit is set (positive difference): diff = v, else v - all
  const uint32_t u = e.y - v;
  const uint32_t m = uint32_t(int32_t(u - v) >> 31);
  if constexpr (DIFF)
    s.acc[K % N] = (e.y & m) - u;
  else
    s.acc[K % N] += (e.y & m) - u;
  s.Pn -= (e.x & 0x800007E0u);
  s.n += 1;
}

#ifdef RSX_LF_PAIRWIN
// Experiment (round 6): ONE window read for two symbols.  The even step reads three dwords of the
// lane's column -- bits [pos, pos + 64) -- and the odd step takes its window out of those: the
// second symbol starts total1 <= 26 bits further and needs at most 26 bits, 52 <= 64.  The odd
// step loses its address, its two LDS reads and one of the two LDS round trips of a symbol;
// the even step pays one more read and one more funnel shift.
template <int N, int K>
__device__ __forceinline__ void lf_step_even(FastState& s, uint32_t vbase) {
  const uint32_t ad = vbase + (s.Pn & ~1023u);
  const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T), d2 = *(lds_u32p)(ad - 4u * LJ_T);
  const uint32_t sh = s.Pn >> 5;
  const uint32_t w = __builtin_amdgcn_alignbit(d0, d1, sh);
  s.wb = __builtin_amdgcn_alignbit(d1, d2, sh);
  s.w = w;
  const lf_u32x2 e = *(lds_u2p)((w >> 19) & 0x1FF8u);
  const uint32_t v = (w >> (e.x & 31u)) & e.y;
  const uint32_t u = e.y - v;
  const uint32_t m = uint32_t(int32_t(u - v) >> 31);
  s.acc[K % N] += (e.y & m) - u;
  s.Pn -= (e.x & 0x800007E0u);
  s.sh = e.x;
  s.n += 1;
}
template <int N, int K>
__device__ __forceinline__ void lf_step_odd(FastState& s) {
  const uint32_t w = __builtin_amdgcn_alignbit(s.w, s.wb, s.sh); // (w : wb) << total1, by 32 - total1 = sh & 31
  const lf_u32x2 e = *(lds_u2p)((w >> 19) & 0x1FF8u);
  const uint32_t v = (w >> (e.x & 31u)) & e.y;
  const uint32_t u = e.y - v;
  const uint32_t m = uint32_t(int32_t(u - v) >> 31);
  s.acc[K % N] += (e.y & m) - u;
  s.Pn -= (e.x & 0x800007E0u);
  s.n += 1;
}
#endif

// Two tables that alternate symbol by symbol (round 4; the kernel's MT instantiations): both
// 10-bit LUTs in the 8 KB the one table takes otherwise, as 4-byte entries
//   bits 0..4 shift | bits 5..10 total (as above) | bits 16..31: 2^SSSS - 1
//   special: 0x0000FFE0 -- "2047 bits": the lane runs far past the end of its slot and stops;
//   an exit offset of 64 and more is what says so afterwards
// table of the even symbols at LDS address 0, of the odd ones at 4096; a lane alternates
// between the two bases starting from the one its entry state names (the steps of a group
// are unrolled: step K of a lane is symbol K of it, so the base is a static choice of two
// registers).  One VALU instruction more than the one-table step (the mask's shift).
constexpr uint32_t LF_MT_SPECIAL = 0x0000FFE0u;
template <int N, int K>
__device__ __forceinline__ void lf_step_mt(FastState& s, uint32_t vbase, uint32_t lut0,
                                           uint32_t lut1) {
  const uint32_t ad = vbase + (s.Pn & ~1023u);
  const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
  const uint32_t w = __builtin_amdgcn_alignbit(d0, d1, s.Pn >> 5);
  uint32_t ea;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ea) : "v"(w >> 20), "s"(0xFFCu), "v"((K & 1) ? lut1 : lut0));
  const uint32_t e = *(lds_u32p)(ea);
  const uint32_t all = e >> 16;
  const uint32_t v = (w >> (e & 31u)) & all;
  const uint32_t u = all - v;
  const uint32_t m = uint32_t(int32_t(u - v) >> 31);
  s.acc[K % N] += (all & m) - u;
  s.Pn -= (e & 0xFFE0u);
  s.n += 1;
}

// A table PER COMPONENT (round 6; the kernel's TM == 2 instantiations, LjStreamDev::fast == 3):
// what DNG writers emit for linear (3-component) images -- one DHT per component,
// AbstractLJpegDecoder.cpp:181-291, LJpegDecompressor.cpp:102-113 -- and any other pattern of
// up to four tables over the N components of an MCU (A B C, A B B, A B C D, A A B B ...).  The
// table of a symbol is tab_of_phase[index mod N]: the phase is part of every parse state
// (offset | phase << 6), as the table bit is for two alternating tables.  N 10-bit LUTs of
// TWO-byte entries in the 8 KB the one 8-byte table takes (2 KB each):
//   bits 0..4 shift (32 - total) | bits 5..10 total (32 * total at bit 0) | bits 11..15 SSSS
//   special: bit 15 set, total 63 (the lane strides on to the end of its slot; the OR of the
//   entries a lane has read says so afterwards)
// 2^SSSS - 1 comes out of the entry with a shift and v_bfm_b32: two VALU instructions a symbol
// more than the two-table step.  Step K of a lane is its K-th symbol (the steps are unrolled),
// so the LUT's base is a static choice among N registers that the lane rotates by its entry
// state's phase once.
constexpr uint32_t LF_PT_SPECIAL = 0x8000u | (63u << 5);
template <int N, int C>
__device__ __forceinline__ void lf_step_pt(FastState& s, uint32_t vbase, uint32_t lut) {
  const uint32_t ad = vbase + (s.Pn & ~1023u);
  const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
  const uint32_t w = __builtin_amdgcn_alignbit(d0, d1, s.Pn >> 5);
  uint32_t ea;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ea) : "v"(w >> 21), "s"(0x7FEu), "v"(lut));
  const uint32_t e = *(const __attribute__((address_space(3))) uint16_t*)(ea);
  uint32_t all;
  asm("v_bfm_b32 %0, %1, 0" : "=v"(all) : "v"(e >> 11));
  const uint32_t v = (w >> (e & 31u)) & all;
  const uint32_t u = all - v;
  const uint32_t m = uint32_t(int32_t(u - v) >> 31);
  s.acc[C] += (all & m) - u;
  s.Pn -= (e & 0x7E0u);
  s.spec |= e;
  s.n += 1;
}

// (KB: symbols of the lane in front of this group of eight -- the component of step K is
// (KB + K) mod N, which is K mod N for 1, 2 and 4 components and not for 3)
template <int N, int K, int KEND, int TM = 0, int KB = 0, bool DIFF = false>
struct LfChain {
  static __device__ __forceinline__ void run(FastState& s, uint32_t vbase, uint32_t pend,
                                             uint32_t (&R)[LF_NR], int qbase, uint32_t lut0,
                                             uint32_t lut1, uint32_t lut2, uint32_t lut3) {
    constexpr int C = (KB + K) % N;
    if (s.Pn > pend) {
      if constexpr (TM == 1)
        lf_step_mt<N, K>(s, vbase, lut0, lut1);
      else if constexpr (TM == 2)
        lf_step_pt<N, C>(s, vbase, C == 0 ? lut0 : (C == 1 ? lut1 : (C == 2 ? lut2 : lut3)));
      else {
#ifdef RSX_LF_PAIRWIN
        if constexpr ((K & 1) == 0)
          lf_step_even<N, C>(s, vbase);
        else
          lf_step_odd<N, C>(s);
#else
        lf_step<N, C, DIFF>(s, vbase);
#endif
      }
      if ((K & 1) == 0)
        s.ev = s.acc[C];
      else
        R[qbase + (K >> 1)] = pack16(s.ev, s.acc[C]);
      if constexpr (K + 1 < KEND)
        LfChain<N, K + 1, KEND, TM, KB, DIFF>::run(s, vbase, pend, R, qbase, lut0, lut1, lut2, lut3);
    }
  }
};

template <int N, int G, int TM = 0, bool DIFF = false>
__device__ __forceinline__ void lf_groups(FastState& s, uint32_t vbase, uint32_t pend,
                                          uint32_t (&R)[LF_NR], uint32_t lut0 = 0,
                                          uint32_t lut1 = 0, uint32_t lut2 = 0, uint32_t lut3 = 0) {
  if (__any(s.Pn > pend)) {
    LfChain<N, 0, 8, TM, 8 * G, DIFF>::run(s, vbase, pend, R, 4 * G, lut0, lut1, lut2, lut3);
    if constexpr (G + 1 < LF_MAXSYM / 8)
      lf_groups<N, G + 1, TM, DIFF>(s, vbase, pend, R, lut0, lut1, lut2, lut3);
  }
}

// lj_slow_entry, inlined: a CALL while 64 VGPRs of running sums are live makes the
// register allocator park half of them in scratch (the ABI's caller-saved registers)
// The symbol entry (code length | SSSS << 5 | total << 10, 0 = invalid code) the general
// way, from the stream's table in GLOBAL memory: the symbols the 10-bit LUT does not
// cover are rare, and the 2.3 KB of LDS the table took (plus the four dependent
// load-store rounds that staged it) are not.
__device__ __forceinline__ uint32_t lf_slow_entry(uint32_t w, const TabLds& tb) {
  // (every load that does not depend on another one asked for at once: the table lies in
  // global memory, and as a loop over the lengths -- look up, compare, next -- a 13-bit code
  // was five dependent round trips, 3-4 us of the lane that re-decodes for a whole workgroup)
  uint32_t r = lj_lut16(tb, w >> (32 - LUT_BITS));
  constexpr int NL = 16 - LUT_BITS;
  uint32_t mc[NL], vo[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    mc[k] = tb.max_code[LUT_BITS + 1 + k];
    vo[k] = tb.val_offset[LUT_BITS + 1 + k];
  }
  const uint32_t max_len = tb.max_len, fix16 = tb.fix16;
  if ((r & 31u) != 0u)
    return r;
  uint32_t len = 0, idx = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const uint32_t l = uint32_t(LUT_BITS + 1 + k), c = w >> (32 - l);
    if (len == 0u && l <= max_len && mc[k] != NO_CODE && c <= mc[k]) {
      len = l;
      idx = (c - vo[k]) & 0xFFFFu;
    }
  }
  if (len == 0u)
    return 0u;
  const uint32_t ssss = tb.values[idx];
  const uint32_t extra = ssss == 16u ? (fix16 ? 16u : 0u) : ssss;
  return len | (ssss << 5) | ((len + extra) << 10);
}

// Re-decode of a slot from a known entry state: the lean step of the fast loop for the
// symbols the 10-bit LUT covers, the general one (any code length, SSSS = 16, invalid
// codes) for the others; the running sum of every symbol goes into the lane's
// side-buffer entry.  (The first version ran the general loop for every symbol: 14-50 us
// per round, and every workgroup behind the re-decoding one waits for its record.)
// x mod P for the phase periods there are (a table per phase: P = 2, 3 or 4, wave-uniform)
__device__ __forceinline__ uint32_t lf_pmod(uint32_t x, uint32_t P) {
  return P == 3u ? x % 3u : (x & (P - 1u));
}
template <int N, int TM, bool DIFF = false>
__device__ __forceinline__ void lf_redecode(const FastLds& F, const TabLds* tabs0, uint32_t tabsel,
                                            uint32_t P, int col, uint32_t start, uint32_t end_bits,
                                            uint32_t side_addr, bool enabled,
                                            uint32_t& exit, uint32_t& count, uint2& sums,
                                            bool& overflow) {
  constexpr bool MT = TM == 1, PT = TM == 2;
  const uint32_t vbase = lds_addr(&F.B[(LF_BW - 1) * LJ_T + col]);
  const uint32_t pend = uint32_t(-32) - 32u * end_bits;
  uint32_t Pn = uint32_t(-32) - 32u * (start & ST_OFF_MASK);
  bool ok = !(start & ST_ERR);
  if (!ok || !enabled)
    Pn = pend; // no steps
  uint32_t n = 0, a0 = 0, a1 = 0, ph3 = 0;
  // (several tables: which one is next -- the table bit of two alternating tables, the phase
  // 0 .. N - 1 of per-component tables; tabsel: the stream's table of phase k in bits 4k..4k+3)
  uint32_t odd = TM != 0 ? ((start >> ST_PHASE_SHIFT) & (MT ? 1u : 3u)) : 0u;
  while (Pn > pend) {
    const uint32_t ad = vbase + (Pn & ~1023u);
    const uint32_t d1 = *(lds_u32p)(ad), d0 = *(lds_u32p)(ad + 4u * LJ_T);
    const uint32_t w = __builtin_amdgcn_alignbit(d0, d1, Pn >> 5);
    lf_u32x2 e;
    if (MT) {
      const uint32_t e4 = *(lds_u32p)(((w >> 20) & 0xFFCu) | (odd ? 4096u : 0u));
      e = lf_u32x2{e4 == LF_MT_SPECIAL ? 0x80000000u : (e4 & 0x7FFu), e4 >> 16};
    } else if (PT) {
      const uint32_t e2 =
          *(const __attribute__((address_space(3))) uint16_t*)(((w >> 21) & 0x7FEu) | (odd << 11));
      e = lf_u32x2{(e2 & 0x8000u) ? 0x80000000u : (e2 & 0x7FFu), (1u << (e2 >> 11)) - 1u};
    } else {
      e = *(lds_u2p)((w >> 19) & 0x1FF8u);
    }
    uint32_t d, tot;
    if (e.x & 0x80000000u) {
      const uint32_t e16 = lf_slow_entry(w, tabs0[(tabsel >> (4u * odd)) & 15u]);
      if (e16 == 0u) {
        ok = false;
        break;
      }
      d = lj_extend(w, e16);
      tot = e16 >> 10;
    } else {
      const uint32_t v = (w >> (e.x & 31u)) & e.y;
      const uint32_t u = e.y - v;
      const uint32_t m = uint32_t(int32_t(u - v) >> 31);
      d = ((e.y & m) - u) & 0xFFFFu;
      tot = (e.x >> 5) & 63u;
    }
    const uint32_t sh = 16u * (n & 1u);
    uint32_t val;
    if (N == 1) {
      a0 = DIFF ? (d & 0xFFFFu) : ((a0 + d) & 0xFFFFu);
      val = a0;
    } else if (N == 2) {
      a0 = pk_add(a0, d << sh);
      val = (a0 >> sh) & 0xFFFFu;
    } else if (N == 3) {
      // (component = n mod 3, kept as a counter; fields 0, 1 in a0, field 2 in a1)
      if (ph3 == 2u) {
        a1 = (a1 + d) & 0xFFFFu;
        val = a1;
      } else {
        a0 = pk_add(a0, d << (16u * ph3));
        val = (a0 >> (16u * ph3)) & 0xFFFFu;
      }
      ph3 = ph3 == 2u ? 0u : ph3 + 1u;
    } else {
      if (n & 2u)
        a1 = pk_add(a1, d << sh);
      else
        a0 = pk_add(a0, d << sh);
      val = (((n & 2u) ? a1 : a0) >> sh) & 0xFFFFu;
    }
    if (n < uint32_t(LF_MAXSYM))
      *(lds_u16w)(side_addr + 2u * n) = uint16_t(val);
    else
      overflow = true;
    Pn -= 32u * tot;
    ++n;
    if (MT)
      odd ^= 1u;
    if (PT)
      odd = odd + 1u == P ? 0u : odd + 1u;
  }
  if (!enabled)
    return;
  exit = ok ? (((pend - Pn) >> 5) | (odd << ST_PHASE_SHIFT)) : ST_ERR;
  count = n;
  sums = make_uint2(a0, a1);
}

// transfer of the predictor state over a run of symbols: T' = f ? Vc + a : T + a,
// Vc' = Vc + v (field-wise; f as 16-bit masks).  NW dwords per quantity (two 16-bit
// fields each): written out per dword so that nothing is carried for absent components.
template <int NW>
struct XferT {
  uint32_t f[NW], a[NW], v[NW];
};
// first h, then g
template <int NW>
__device__ __forceinline__ XferT<NW> xfer_compose(const XferT<NW>& h, const XferT<NW>& g) {
  XferT<NW> r;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    r.f[k] = h.f[k] | g.f[k];
    r.a[k] = pk_add(g.a[k], (h.v[k] & g.f[k]) | (h.a[k] & ~g.f[k]));
    r.v[k] = pk_add(h.v[k], g.v[k]);
  }
  return r;
}
__device__ __forceinline__ uint32_t fld_mask1(uint32_t flags2) { // two flag bits -> two 16-bit masks
  return ((flags2 & 1u) ? 0xFFFFu : 0u) | ((flags2 & 2u) ? 0xFFFF0000u : 0u);
}

// The window of a look-back-1 pass: LF_LB1_WAVES wavefronts of 64 records.  256 records until
// round 5 -- "whatever the stream has in flight" --; but a pass asks for 32-64 B of EVERY
// record in its window, a line each, and the nearest inclusive state is a handful of
// workgroups away: with one wavefront's 64 the kernel is 3 % (cfg 3), 8 % (cfg 4), 15 %
// (4 frames of noise; 3 components) faster, and a stream alone on the chip (1024 workgroups
// in flight) still gains 6 % (profiles/r05/ab_lookback_window.txt).  The first pass looks at
// the LF_LB1_WIN0 nearest only: 16 or 32 are another 2-5 % over 64, 8 and 4 no better.
#ifndef LF_LB1_WAVES
#define LF_LB1_WAVES 1
#endif
#ifndef LF_LB1_WIN0
#define LF_LB1_WIN0 16
#endif
// Look-back 1 (the whole workgroup, one record per lane and pass): the predictor state
// (T, Vc) before the workgroup = the nearest inclusive state, carried through the LOCAL
// transfers of the workgroups in between.  The transfers compose associatively: a
// wavefront scan (nearer workgroups applied later), the four wavefronts' results in LDS.
// (First version: one wavefront, a serial fold over its 64 lanes -- 10 us a workgroup.)
template <int N, int lbw>
__device__ __forceinline__ bool lb1_walk(const LjArgs& a, const FastLds& F, uint32_t b,
                                         uint32_t first_block, uint2 init, int j, uint2* T_in,
                                         uint2* V_in) {
  constexpr int NW = (N + 1) / 2;
  const int lane = j & 63, wv = j >> 6;
  const u64* A = a.lb;
  XferT<NW> g; // the workgroups nearer than the window, composed
#pragma unroll
  for (int k = 0; k < NW; ++k)
    g.f[k] = g.a[k] = g.v[k] = 0;
  const uint32_t initw[2] = {init.x, init.y};
  int64_t pos = int64_t(b) - 1;
  // (the records a pass asks for: the first pass of the one-wavefront walk looks at the
  // LF_LB1_WIN0 nearest only, the passes behind it -- all of them LOCAL -- at 64)
  int win = lbw == 1 ? LF_LB1_WIN0 : 64;
  for (uint32_t spins = 0; spins < LF_SPIN_LIMIT; ++spins) {
    const int64_t idx = pos - j;
    const bool inwin = wv < lbw && lane < win; // (outside the window: no record, LOCAL nothing)
    const bool real = idx >= int64_t(first_block);
    u64 wa[NW], wv_[NW], wt[NW], wc[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k)
      wa[k] = wv_[k] = wt[k] = wc[k] = 0;
    if (real && inwin) {
      const u64* p = A + size_t(idx) * LF_LB_WORDS;
#ifdef RSX_LF_LB16
      // (coherent 16-byte loads: agent scope is `sc1`, as the compiler emits it for lb_load)
      lf_u32x4 q[2 * NW];
      if constexpr (NW == 1)
        asm volatile("global_load_dwordx4 %0, %2, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %1, %2, off offset:32 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1])
                     : "v"(p)
                     : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %4, off offset:16 sc1\n\t"
                     "global_load_dwordx4 %1, %4, off offset:32 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:48 sc1\n\t"
                     "global_load_dwordx4 %3, %4, off offset:64 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2 * NW - 2]), "=&v"(q[2 * NW - 1])
                     : "v"(p)
                     : "memory");
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        wa[k] = u64(q[2 * k][0]) | (u64(q[2 * k][1]) << 32);
        wv_[k] = u64(q[2 * k][2]) | (u64(q[2 * k][3]) << 32);
        wt[k] = u64(q[2 * k + 1][0]) | (u64(q[2 * k + 1][1]) << 32);
        wc[k] = u64(q[2 * k + 1][2]) | (u64(q[2 * k + 1][3]) << 32);
      }
#else
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        wa[k] = lb_load(p + lb_wa(k));
        wv_[k] = lb_load(p + lb_wv(k));
        wt[k] = lb_load(p + lb_wt(k));
        wc[k] = lb_load(p + lb_wc(k));
      }
#endif
    }
    bool loc = true, pre = true;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      loc = loc && (wa[k] & LB_VALID) && (wv_[k] & LB_VALID);
      pre = pre && (wt[k] & LB_VALID) && (wc[k] & LB_VALID);
    }
    if (!real) { // before the stream: the initial predictors
      pre = true;
      loc = false;
    }
    const u64 m_pre = __ballot(pre && inwin), m_loc = __ballot(loc && !pre);
    // lanes 0 .. f-1 LOCAL, lane f inclusive state (f == win: none in the window)
    const int f = m_pre ? __builtin_ctzll(m_pre) : win;
    const u64 need = f == 64 ? ~0ull : ((1ull << f) - 1ull);
    const bool wave_ok = (m_loc & need) == need;
    // inclusive scan of the transfers: lane t <- lanes 0..t, lane 0 (nearest) applied last
    XferT<NW> c;
    const uint32_t fl = uint32_t(wa[0] >> 32) & 0xFu;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      c.a[k] = uint32_t(wa[k]);
      c.v[k] = uint32_t(wv_[k]);
      c.f[k] = fld_mask1(fl >> (2 * k));
    }
    // (DPP network; a lane without a source composes with the identity transfer 0 / 0 / 0)
#define LF_STEP_XFER(C, M)                         \
  {                                                \
    XferT<NW> nr; /* the nearer lanes' segment */  \
    _Pragma("unroll") for (int k = 0; k < NW; ++k) { \
      nr.a[k] = dpp0<C, M>(c.a[k]);                \
      nr.v[k] = dpp0<C, M>(c.v[k]);                \
      nr.f[k] = dpp0<C, M>(c.f[k]);                \
    }                                              \
    c = xfer_compose<NW>(c, nr); /* own (farther) first, then the nearer ones */ \
  }
    LF_DPP_SCAN(LF_STEP_XFER)
#undef LF_STEP_XFER
    // summary of the wavefront: [0] f, [1] ok, [2..7] transfer of lanes 0..f-1,
    // [8..11] state of lane f
    uint32_t* X = F.misc + M_LBX + 12 * wv;
    if (lane == 0) {
      X[0] = uint32_t(f);
      X[1] = wave_ok ? 1u : 0u;
      if (f == 0) { // (no LOCAL lanes in front of the state: identity)
#pragma unroll
        for (int k = 0; k < 6; ++k)
          X[2 + k] = 0u;
      }
    }
    if (f > 0 && lane == f - 1) {
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        X[2 + k] = c.f[k];
        X[4 + k] = c.a[k];
        X[6 + k] = c.v[k];
      }
    }
    if (f < win && lane == f) {
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        X[8 + k] = real ? uint32_t(wt[k]) : initw[k];
        X[10 + k] = real ? uint32_t(wc[k]) : initw[k];
      }
    }
    lds_barrier();
    int state = 0; // 0: all 256 LOCAL, 1: found an inclusive state, 2: blocked
    uint32_t Tf[NW], Vf[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k)
      Tf[k] = Vf[k] = 0;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      if (state != 0 || w4 >= lbw)
        continue;
      const uint32_t* Y = F.misc + M_LBX + 12 * w4;
      if (!uni(Y[1])) {
        state = 2;
        continue;
      }
      XferT<NW> h;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        h.f[k] = uni(Y[2 + k]);
        h.a[k] = uni(Y[4 + k]);
        h.v[k] = uni(Y[6 + k]);
      }
      g = xfer_compose<NW>(h, g); // this wavefront's lanes are farther than everything so far
      if (uni(Y[0]) < uint32_t(win)) {
        state = 1;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          Tf[k] = uni(Y[8 + k]);
          Vf[k] = uni(Y[10 + k]);
        }
      }
    }
    lds_barrier(); // (the exchange words are free again)
    if (state == 1) {
      uint32_t to[2] = {0, 0}, vo[2] = {0, 0};
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        to[k] = pk_add(g.a[k], (Vf[k] & g.f[k]) | (Tf[k] & ~g.f[k]));
        vo[k] = pk_add(Vf[k], g.v[k]);
      }
      *T_in = make_uint2(to[0], to[1]);
      *V_in = make_uint2(vo[0], vo[1]);
      return true;
    }
    if (state == 0) {
      pos -= lbw == 1 ? win : 64 * lbw;
      win = 64;
      continue;
    }
    // blocked: forget this pass's partial composition and look again
#pragma unroll
    for (int k = 0; k < NW; ++k)
      g.f[k] = g.a[k] = g.v[k] = 0;
    pos = int64_t(b) - 1;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}

// The stream's fields the kernel uses, read ONCE and made wave-uniform.  Through a
// reference to a.streams[s] -- s itself a loaded value -- the compiler takes every field
// for a per-lane value: a vector load plus a wait at every use, re-issued after every
// store (measured: 16 us of a workgroup's 55 in the copy-out alone).
struct FastStream {
  uint32_t fast_n; // fast ? direct : 0
  uint32_t tm;     // 0 one table, 1 two tables alternating symbol by symbol, 2 a table per phase
  uint32_t tabsel; // the stream's table of phase k (symbol index mod N) in bits 4k .. 4k + 3
  uint32_t tp;     // tm == 2: the period of that assignment (2, 3, 4): a state's phase is mod this
  uint32_t diffs;  // the stream leaves differences for the legacy reconstruction (fast_diffs)
  uint32_t nk;     // a Nikon-type stream whose pixels this kernel writes (fast_nk)
  uint64_t diff_offset;
  uint32_t first_block, first_subseq, table_base, start_bit, n_blocks;
  uint32_t RS, kind, keep, out_x, out_y, pitch, n_strips, strip_base;
  uint64_t needed, img_offset;
  uint2 init;
};
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
  return uint64_t(uni(uint32_t(x))) | (uint64_t(uni(uint32_t(x >> 32))) << 32);
}
__device__ __forceinline__ FastStream lf_stream(const LjStreamDev& S) {
  FastStream f;
  // (3 components, round 5: not a stream of the fused multi-kernel path -- direct == 0 --, so
  // the number of components says which instantiation takes it)
  f.diffs = uni(S.fast && S.fast_diffs ? 1u : 0u);
  f.nk = uni(S.fast && S.fast_nk ? 1u : 0u);
  f.diff_offset = uni64(S.diff_offset);
  // (a stream that leaves differences is one "component": its symbols in stream order; a Nikon-type
  // row is two interleaved ones, the columns' parities)
  f.fast_n = uni(S.fast ? (f.diffs ? 1u : (f.nk ? 2u : (S.direct ? uint32_t(S.direct) : S.n_comp))) : 0u);
  f.tm = uni(S.fast >= 2 ? uint32_t(S.fast) - 1u : 0u);
  f.tabsel = uni(f.tm ? (uint32_t(S.tab_of_phase[0] & 15u) | (uint32_t(S.tab_of_phase[1] & 15u) << 4) |
                         (uint32_t(S.tab_of_phase[2] & 15u) << 8) |
                         (uint32_t(S.tab_of_phase[3] & 15u) << 12))
                      : 0u);
  f.tp = uni(f.tm == 2u ? uint32_t(S.tab_period) : 1u);
  f.first_block = uni(S.first_block);
  f.first_subseq = uni(S.first_subseq);
  f.table_base = uni(S.table_base);
  f.start_bit = uni(S.start_bit);
  f.n_blocks = uni(S.n_blocks);
  f.RS = uni(S.row_samples);
  f.kind = uni(S.kind);
  f.keep = uni(S.keep_samples < S.row_samples ? S.keep_samples : S.row_samples);
  f.out_x = uni(S.out_x);
  f.out_y = uni(S.out_y);
  f.pitch = uni(S.img_pitch);
  f.n_strips = uni(S.n_strips);
  f.strip_base = uni(S.strip_base);
  f.needed = uni64(S.needed);
  f.img_offset = uni64(S.img_offset);
  f.init = make_uint2(uni(uint32_t(S.init_pred[0]) | (uint32_t(S.init_pred[1]) << 16)),
                      uni(uint32_t(S.init_pred[2]) | (uint32_t(S.init_pred[3]) << 16)));
  return f;
}

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
// Copy-out: the staged running sums of stream symbols [A0, A1) (LDS, stream order from
// byte address sb) -> image.  The samples fall into RUNS that are contiguous in the image
// (a stream row's kept part, a CR2 strip row); a run is written as 16-byte chunks on the
// destination's 16-byte grid (partial chunks at its ends sample by sample).
// (Round 3 dealt CHUNKS to lanes and had every lane walk the run list for itself: with runs
// of ~280 chunks and a stride of 256 nearly every chunk of a lane lay in another run than
// its last one, so the run's set-up -- address arithmetic in 64 bits, the row's constants,
// the CR2 strip: ~100 vector instructions -- was paid per 16 bytes, a third of the kernel's
// vector instructions.)  The run list is walked ONCE PER WAVEFRONT with wave-uniform state
// (scalar registers, the scalar unit), every run is cut into segments of 64 chunks, and the
// segments are dealt to the four wavefronts round robin; a lane's work per chunk is what
// only it can do: five LDS reads, four funnel shifts, four packed adds, one 16-byte store.
// ---------------------------------------------------------------------------
// where the walk over the runs starts (CR2: the strip of the workgroup's first sample and
// the place in it: a search over the strips and a division) -- worked out BEFORE look-back 1,
// in the shadow of its wait, not behind it
struct WalkStart {
  uint32_t z, srow, col, sw, sx0, sy0, znext;
};
__device__ __forceinline__ WalkStart lf_walk_start(const FastLds& F, const FastStream& S,
                                                   uint32_t A0_) {
  WalkStart w{0, 0, 0, 1, 0, 0, 0xFFFFFFFFu};
  if (S.kind == 1) {
    const uint32_t i = uni(A0_);
    const Cr2Strip* st = reinterpret_cast<const Cr2Strip*>(F.strips);
    uint32_t z = 0;
    while (z + 1 < S.n_strips && uint64_t(i) >= uni64(st[z + 1].first_sample))
      ++z;
    w.z = z;
    w.sx0 = uni(st[z].x0);
    w.sw = uni(st[z].w);
    w.sy0 = uni(st[z].y0);
    uint32_t srow, col;
    strip_divmod(uint64_t(i) - uni64(st[z].first_sample), w.sw, &srow, &col);
    w.srow = uni(srow);
    w.col = uni(col);
    w.znext = z + 1 < S.n_strips ? uint32_t(uni64(st[z + 1].first_sample)) : 0xFFFFFFFFu;
  }
  return w;
}

// What the copy-out of a Nikon-type stream (fast_nk) does to a value on its way out
// (NikonDecompressor.cpp:518-560: rawdata->setWithLookUp(clampBits(pLeft, 15), dest, &random);
// PentaxDecompressor.cpp:155-177: the value must fit the sensor's bits, stored as it is).
struct NkOut {
  uint32_t dither;      // the curve with its dither (TableLookUp's dither form, common/RawImage.h:335-353)
  uint32_t limit_shift; // a value v with v >> limit_shift != 0 gives the stream to the legacy route
  uint32_t seed;        // the 24 bits at the stream's start: the dither generator's first state
  const uint32_t* tab;    // 32768 x (base | delta << 16)
  const uint32_t* rowpow; // 15700^(pixels in front of stream row r) mod m
  const uint32_t* colpow; // 15700^x mod m, x < RS
  uint32_t* flags;        // the stream's LjResult::flags
};
// (the generator is a lag-1 multiply-with-carry: r' = 15700 (r & 65535) + (r >> 16), i.e.
// r_n = r_0 15700^n mod m, m = 15700 * 2^16 - 1 -- rsx_ljpeg_recon.hip has the argument)
__device__ __forceinline__ uint32_t lf_nk_mulmod(uint32_t x, uint32_t y) {
  return uint32_t((uint64_t(x) * y) % (15700ull * 65536ull - 1ull));
}

template <int N, bool NK = false>
__device__ __forceinline__ void lf_copy_out2(const FastLds& F, const LjArgs& a,
                                             const FastStream& S, uint32_t A0_, uint32_t A1_,
                                             uint32_t sb, uint32_t r0_, int tid,
                                             const WalkStart& ws, uint8_t* img,
                                             const NkOut& nk = NkOut{}) {
  const uint32_t lane = uint32_t(tid) & 63u;
  const uint32_t wv = uni(uint32_t(tid) >> 6);
  // (every lane holds the same values: said so, or the walk below runs on vector registers)
  const uint32_t A0 = uni(A0_), A1 = uni(A1_), r0 = uni(r0_);
  const uint32_t RS = S.RS;
  const Cr2Strip* st = reinterpret_cast<const Cr2Strip*>(F.strips);
  // cursor (wave-uniform): the run that starts at sample i
  uint32_t i = A0, r = r0, sidx = A0 - r0 * RS;
  uint32_t z = uni(ws.z), srow = uni(ws.srow), col = uni(ws.col), sw = uni(ws.sw),
           sx0 = uni(ws.sx0), sy0 = uni(ws.sy0), znext = uni(ws.znext);
  // (the row's constants: read when the row changes, not per run)
  uint2 C;
  {
    const uint2 Cv = F.ctab[0];
    C = make_uint2(uni(Cv.x), uni(Cv.y));
  }
  uint32_t gseg = 0; // segments dealt so far
  uint32_t rowst = 0; // (Nikon-type with dither: the generator's state at the row's first pixel)
  if constexpr (NK)
    if (nk.dither)
      rowst = uni(lf_nk_mulmod(uni(nk.seed), nk.rowpow[r]));
  while (i < A1) {
    uint32_t n;
    uint8_t* dst = nullptr;
    if (S.kind == 0) {
      if (sidx < S.keep) {
        n = S.keep - sidx;
        dst = img + uint64_t(S.out_y + r) * S.pitch + 2u * (S.out_x + sidx);
      } else {
        n = RS - sidx; // trailing MCUs of the frame that the tile does not keep
      }
    } else {
      const uint32_t in_strip = sw - col, in_row = RS - sidx;
      n = in_strip < in_row ? in_strip : in_row;
      dst = img + uint64_t(sy0 + srow) * S.pitch + 2u * (sx0 + col);
    }
    if (n > A1 - i)
      n = A1 - i;
    if (dst) {
      const uint32_t delta = uint32_t(reinterpret_cast<uintptr_t>(dst) & 15u) >> 1;
      // (the component of a chunk's first sample: i + 8 m - delta mod N -- the same for every
      // chunk m of the run while 8 is a multiple of N; N = 3: three sets of constants, the
      // chunk takes the one of its m mod 3)
      const uint32_t ph0 = lf_mod<N>(i + (N == 3 ? 9u : 8u) - delta); // (delta <= 7)
      uint32_t cd[4], cd1[4], cd2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        cd[t] = fld(C, lf_mod<N>(ph0 + 2u * t)) | (fld(C, lf_mod<N>(ph0 + 2u * t + 1u)) << 16);
        if (N == 3) { // (8 m mod 3 = 2 m mod 3: m mod 3 = 1 shifts the phase by 2, = 2 by 1)
          cd1[t] = fld(C, lf_mod<N>(ph0 + 2u + 2u * t)) | (fld(C, lf_mod<N>(ph0 + 3u + 2u * t)) << 16);
          cd2[t] = fld(C, lf_mod<N>(ph0 + 1u + 2u * t)) | (fld(C, lf_mod<N>(ph0 + 2u + 2u * t)) << 16);
        }
      }
      const uint32_t nch = (delta + n + 7u) >> 3;
      const uint32_t lds0 = sb + 2u * (i - A0) - 2u * delta;
      const uint32_t sh = 8u * (lds0 & 2u); // the staged samples start mid-dword: 16, else 0
      uint8_t* const d0 = dst - 2u * delta;
      const uint32_t nseg = (nch + 63u) >> 6;
      // chunk m of the run: its eight samples, the row's constants added
      auto chunk = [&](uint32_t m, uint32_t (&o)[4]) {
        const uint32_t la = (lds0 + 16u * m) & ~3u;
        uint32_t dw[5];
#pragma unroll
        for (int t = 0; t < 5; ++t)
          dw[t] = *(lds_u32p)(la + 4u * t);
        const uint32_t m3 = N == 3 ? m % 3u : 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t cc = N == 3 ? (m3 == 0u ? cd[t] : (m3 == 1u ? cd1[t] : cd2[t])) : cd[t];
          o[t] = pk_add(__builtin_amdgcn_alignbit(dw[t + 1], dw[t], sh), cc);
        }
      };
      // ... written out (whole, or sample by sample at a run's ends)
      auto put = [&](uint32_t m, const uint32_t (&o)[4]) {
        const int32_t sf = int32_t(8u * m) - int32_t(delta);
        uint8_t* p = d0 + 16u * m;
        if (LF_ABLATE & 256u) { // (experiment: everything but the stores)
          asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(p));
        } else if (sf >= 0 && uint32_t(sf) + 8u <= n) {
#ifndef RSX_LF_PLAIN_MEM
          __builtin_nontemporal_store(lf_u32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<lf_u32x4*>(p));
#else
          *reinterpret_cast<uint4*>(p) = make_uint4(o[0], o[1], o[2], o[3]);
#endif
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int32_t q = sf + t;
            if (q >= 0 && uint32_t(q) < n)
              reinterpret_cast<uint16_t*>(p)[t] = uint16_t(o[t >> 1] >> (16 * (t & 1)));
          }
        }
      };
      // Nikon-type: is a value of the chunk outside what the mod-2^16 sums can vouch for?  (The sums
      // are mod 2^16, the reference's are ints: as long as every value so far was inside
      // 0 .. 2^limit_shift - 1 <= 32767 and a difference is at most 2^15 in size, the next one is
      // outside as an int exactly if it is outside mod 2^16.)  A run's first and last chunk: the
      // samples qa <= t < qb belong to the run -- what lies next to them in the staging region is
      // another row's, or nobody's.
      auto nk_check = [&](uint32_t m, const uint32_t (&o)[4]) {
        const uint32_t over = ((0xFFFFu << nk.limit_shift) & 0xFFFFu) * 0x10001u;
        const int32_t sf = int32_t(8u * m) - int32_t(delta);
        const int32_t qa = sf < 0 ? -sf : 0;
        const int32_t qb = int32_t(n) - sf < 8 ? int32_t(n) - sf : 8;
        uint32_t bad = (o[0] | o[1] | o[2] | o[3]) & over;
        if (qa != 0 || qb != 8) {
          bad = 0;
#pragma unroll
          for (int t = 0; t < 8; ++t)
            bad |= (t >= qa && t < qb) ? (o[t >> 1] >> (16 * (t & 1))) & over & 0xFFFFu : 0u;
        }
        if (bad)
          atomicOr(nk.flags, FL_SLOW);
      };
      // ... the curve's entries of its eight values and the generator's power of its first column,
      // asked for at once (no load behind a branch).  The generator's state belongs to the pixel's
      // place in decode order, so the state of the chunk's sample 0 is right whether or not the
      // samples in front of qa are the run's -- they are this row's (a run that starts mid-row), and
      // what is computed for samples outside the run is not stored.  Only a chunk that begins in
      // front of the row's first pixel (a row that does not start on the 16-byte grid) starts at qa.
      auto nk_ask = [&](uint32_t m, const uint32_t (&o)[4], uint32_t (&e)[8], uint32_t& cp, int32_t& first) {
        const int32_t sf = int32_t(8u * m) - int32_t(delta);
        const int32_t x0 = int32_t(sidx) + sf;
        first = x0 >= 0 ? 0 : -sf;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          e[t] = nk.tab[(o[t >> 1] >> (16 * (t & 1))) & 0x7FFFu];
        cp = nk.colpow[uint32_t(x0 + first)];
      };
      // ... and the values on their way out (TableLookUp's dither form, common/RawImage.h:335-353)
      auto nk_dither = [&](uint32_t (&o)[4], const uint32_t (&e)[8], uint32_t cp, int32_t first) {
        uint32_t st = lf_nk_mulmod(rowst, cp);
        uint32_t v[8];
        if (first == 0) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            v[t] = ((e[t] & 0xFFFFu) + (((e[t] >> 16) * (st & 2047u) + 1024u) >> 12)) & 0xFFFFu;
            st = 15700u * (st & 65535u) + (st >> 16);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            v[t] = ((e[t] & 0xFFFFu) + (((e[t] >> 16) * (st & 2047u) + 1024u) >> 12)) & 0xFFFFu;
            st = t >= first ? 15700u * (st & 65535u) + (st >> 16) : st;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          o[t] = v[2 * t] | (v[2 * t + 1] << 16);
      };
      // (Tried: two of a wavefront's segments at a time, the second one's table entries on their way
      // while the first one's values are worked out -- 9.4 us a workgroup against 8.6: the entries
      // are not waited for one after the other as it is; what the curve costs is the 64 lanes' 64
      // places in a 128 KB table, a cache line each, eight times a chunk.)
      for (uint32_t seg = (wv - gseg) & 3u; seg < nseg; seg += 4u) {
        const uint32_t m = seg * 64u + lane;
        if (m < nch) {
          uint32_t o[4];
          chunk(m, o);
          if constexpr (NK) {
            nk_check(m, o);
            if (nk.dither) {
              uint32_t e[8], cp;
              int32_t first;
              nk_ask(m, o, e, cp, first);
              nk_dither(o, e, cp, first);
            }
          }
          put(m, o);
        }
      }
      gseg += nseg;
    }
    // move the cursor past the run
    i += n;
    sidx += n;
    if (sidx == RS) {
      sidx = 0;
      ++r;
      if (i < A1) {
        const uint2 Cv = F.ctab[r - r0];
        C = make_uint2(uni(Cv.x), uni(Cv.y));
        if constexpr (NK)
          if (nk.dither)
            rowst = uni(lf_nk_mulmod(uni(nk.seed), nk.rowpow[r]));
      }
    }
    if (S.kind == 1) {
      col += n;
      if (col == sw) {
        col = 0;
        ++srow;
      }
      if (i >= znext && i < A1) {
        ++z;
        sx0 = uni(st[z].x0);
        sw = uni(st[z].w);
        sy0 = uni(st[z].y0);
        srow = 0;
        col = 0;
        znext = z + 1 < S.n_strips ? uint32_t(uni64(st[z + 1].first_sample)) : 0xFFFFFFFFu;
      }
    }
  }
}

// Staging of a lane's register pairs q < nq (static register indices; groups of four
// pairs are skipped wave-uniformly once nobody has any left).  Round 4: the four pairs of
// a group in one hand-written block.  Left to the compiler every pair was "compare, save
// exec, branch, add, two writes, restore exec" with the branches taken or not lane set by
// lane set -- 2 us a workgroup whether four lanes staged or all of them (the two-phase
// experiment, profiles/r04).  The lanes that still have pair q are a SUBSET of those that
// had pair q - 1, so v_cmpx (it writes EXEC) only ever narrows the set: no branch, no
// restore inside the group; 18 instructions for four pairs.
template <int N, int Q4>
__device__ __forceinline__ void lf_stage(const uint32_t (&R)[LF_NR], uint32_t ad, uint32_t nq,
                                         uint32_t nqmax, uint32_t kk0, uint32_t kk1,
                                         uint32_t kk2) {
  if (uint32_t(4 * Q4) < nqmax) { // (wave-uniform)
    uint64_t saved;
    uint32_t t;
    // the constants of pairs 4 Q4 .. 4 Q4 + 3 (static choices: see the caller)
    const uint32_t kq[3] = {kk0, kk1, kk2};
    const uint32_t k0 = N == 3 ? kq[(4 * Q4) % 3] : kk0, k1 = N == 3 ? kq[(4 * Q4 + 1) % 3] : kk1,
                   k2 = N == 3 ? kq[(4 * Q4 + 2) % 3] : kk0, k3 = N == 3 ? kq[(4 * Q4 + 3) % 3] : kk1;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_u32_e32 vcc, %[q0], %[nq]\n\t"
                 "v_pk_add_u16 %[t], %[r0], %[k0]\n\t"
                 "ds_write_b16 %[ad], %[t] offset:%[o0]\n\t"
                 "ds_write_b16_d16_hi %[ad], %[t] offset:%[o0h]\n\t"
                 "v_cmpx_lt_u32_e32 vcc, %[q1], %[nq]\n\t"
                 "v_pk_add_u16 %[t], %[r1], %[k1]\n\t"
                 "ds_write_b16 %[ad], %[t] offset:%[o1]\n\t"
                 "ds_write_b16_d16_hi %[ad], %[t] offset:%[o1h]\n\t"
                 "v_cmpx_lt_u32_e32 vcc, %[q2], %[nq]\n\t"
                 "v_pk_add_u16 %[t], %[r2], %[k2]\n\t"
                 "ds_write_b16 %[ad], %[t] offset:%[o2]\n\t"
                 "ds_write_b16_d16_hi %[ad], %[t] offset:%[o2h]\n\t"
                 "v_cmpx_lt_u32_e32 vcc, %[q3], %[nq]\n\t"
                 "v_pk_add_u16 %[t], %[r3], %[k3]\n\t"
                 "ds_write_b16 %[ad], %[t] offset:%[o3]\n\t"
                 "ds_write_b16_d16_hi %[ad], %[t] offset:%[o3h]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(saved), [t] "=&v"(t)
                 : [nq] "v"(nq), [ad] "v"(ad), [k0] "v"(k0), [k1] "v"(k1), [k2] "v"(k2),
                   [k3] "v"(k3), [r0] "v"(R[4 * Q4]),
                   [r1] "v"(R[4 * Q4 + 1]), [r2] "v"(R[4 * Q4 + 2]), [r3] "v"(R[4 * Q4 + 3]),
                   [q0] "n"(4 * Q4), [q1] "n"(4 * Q4 + 1), [q2] "n"(4 * Q4 + 2),
                   [q3] "n"(4 * Q4 + 3), [o0] "n"(16 * Q4), [o0h] "n"(16 * Q4 + 2),
                   [o1] "n"(16 * Q4 + 4), [o1h] "n"(16 * Q4 + 6), [o2] "n"(16 * Q4 + 8),
                   [o2h] "n"(16 * Q4 + 10), [o3] "n"(16 * Q4 + 12), [o3h] "n"(16 * Q4 + 14)
                 : "vcc", "memory");
    if constexpr (Q4 + 1 < LF_NR / 4)
      lf_stage<N, Q4 + 1>(R, ad, nq, nqmax, kk0, kk1, kk2);
  }
}

// phase time stamps of experiment builds (scripts/exp_lj_stats.py prints their means)
#ifdef RSX_EXPERIMENT
#define LF_STAMP(k)                                                      \
  do {                                                                   \
    if (a.dbg && j == 0)                                                 \
      a.dbg[size_t(b) * 16 + (k)] = __builtin_amdgcn_s_memtime();        \
  } while (0)
#else
#define LF_STAMP(k) \
  do {              \
  } while (0)
#endif

// ---------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------
// MODE 0: the steady state.  1 (PROBE): a plan's first run, when every LDS level is launched
// and one works.  2: plans laid out on the device (restart intervals) -- PROBE's early look at
// the level, and every wavefront drops the scalar cache before its first load (lj_fresh_scalars)
// DIFF (round 6): the stream's reconstruction stays with the legacy kernels (Nikon-type predictors
// with their curve and dither, Pentax, Canon sRaw groups): the kernel decodes every symbol ONCE, as
// for the others, and leaves the DIFFERENCES in stream order where lj_decode_kernel would have --
// no rows, no look-back 1, one contiguous run to write -- in place of the multi-kernel pipeline's
// warm-up + recorded pass + stitch passes + final decode of such streams.
// NK (round 6): a Nikon-type stream's PIXELS (fast_nk).  Its row is two interleaved components (the
// columns' parities, pLeft1 / pLeft2), but the first pair of a row is predicted from the row TWO
// above (pUp1 / pUp2 by the row's parity): the vertical sums Vc have FOUR fields -- .x the two
// components of the even stream rows, .y of the odd ones -- and so has the open row's state T, of
// which a workgroup reads the half of the row that is open when it starts (the other half carries
// whatever the transfers add to it: nobody looks).  A transfer stays field-wise: a component that
// starts a row here sets the field of THAT row's parity from the same field of Vc.
template <int N, int TM, int MODE, bool DIFF = false, bool NK = false>
__global__ __launch_bounds__(LJ_T, LF_WG_PER_CU) void lj_fast_kernel(LjArgs a, uint32_t lds_bytes,
                                                          uint32_t level) {
  constexpr bool PROBE = MODE >= 1, INV = MODE == 2;
  static_assert(!DIFF || (N == 1 && TM == 0), "differences: one table, symbols in stream order");
  static_assert(!NK || (N == 2 && TM == 0 && !DIFF), "Nikon-type pixels: one table, two column parities");
  constexpr bool MT = TM == 1, PT = TM == 2; // two alternating tables / a table per phase
  constexpr uint32_t TICKET0 = (MT ? 12u : 0u) + (N == 4 ? 2u : (N == 3 ? 3u : uint32_t(N) - 1u));
  // offset (| table bit of the next symbol | its phase, two bits)
  constexpr uint32_t SMASK = MT ? 0x7Fu : (PT ? 0xFFu : ST_OFF_MASK);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const FastLds F = carve_fast(smem, lds_bytes);
  const int j = threadIdx.x, lane = j & 63, wv = j >> 6;
#ifdef RSX_EXPERIMENT
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  // Is this launch the one that works (LjArgs::fast_lds_lv)?  An L2 load: the word changes
  // from run to run, and nothing invalidates a CU's scalar cache between two kernels of a
  // stream (measured: stale level words).  At level 0, the usual one, the ticket is taken
  // in the same round trip (every level has tickets of its own, on another cache line --
  // 15 000 atomics on one address take 0.17 ms, so the other levels ask first).
#ifndef RSX_LF_TICKETS
  // Round 5: the block index IS the ticket.  The dispatcher starts a 1-D grid's workgroups
  // in order (lj_unstuff_kernel's hand-over relies on the same), so a predecessor in
  // fast_order's sequence has started when its successor runs; an atomic ticket + its LDS
  // broadcast + barrier stood in front of everything else a workgroup asks for, a dependent
  // round trip of 1-2 us (and a queue of 1024 on one address at the kernel's start).  Every
  // wait below is bounded, so an order the dispatcher did not keep costs time -- the stream
  // goes to the multi-kernel pipeline --, never a hang or a pixel; the host keeps two LARGE
  // single-pass launches of a context from running side by side (two grids that both fill
  // the chip could each hold the slots the other's next workgroup needs: ljpeg_plan_run).
  // Plans whose streams all have the same number of workgroups (frames of a batch, the tiles
  // of a DNG) work (stream, block) out arithmetically: nothing stands in front of the image
  // loads; the others read fast_order's entry first.
  (void)TICKET0;
  lj_fresh_scalars<INV>();
  const uint32_t t_blk = a.blk0 + blockIdx.x;
  const uint32_t chosen_now = __hip_atomic_load(&a.fast_level[a.run_parity], __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
  // (A plan's FIRST run launches every LDS level it might need and one of them works -- the
  // PROBE instantiation: the others ask for nothing else.  With the level looked at behind
  // the head's loads, as the runs after it do, each of them read the whole un-stuffed image,
  // 2 x 300 MB on cfg 3; as a run-time condition in the one instantiation the test cost every
  // run 1.5-2 %: scripts/rounds/r05/r05q.sh.)
  if constexpr (PROBE)
    if (uni(chosen_now) != level)
      return;
  uint32_t b, s, fb_now, tz_now = 0, tbv = 0, tpv = 0;
  const bool uniform_plan = a.fast_uniform_nb != 0u;
  if (uniform_plan) {
    const uint32_t ns = a.n_streams, nb = a.fast_uniform_nb;
    const uint32_t k = t_blk / ns, i = t_blk - k * ns;
    s = a.fast_rotate ? (i + k % ns) % ns : i;
    b = s * nb + k;
    fb_now = s * nb;
    // (the stream's tables: asked for now, looked at behind the image loads)
    tbv = a.streams[s].table_base;
    tpv = *reinterpret_cast<const uint32_t*>(a.streams[s].tab_of_phase); // (phases 0..3, a byte each)
  } else {
    const uint4 bs = a.fast_order[t_blk];
    b = uni(bs.x);
    s = uni(bs.y);
    tz_now = uni(bs.z);
    fb_now = uni(bs.w);
    if (s == 0xFFFFFFFFu)
      return; // (a block no stream owns: plans laid out on the device, lj_dri_layout_kernel)
  }
#else
  if (j == 0) {
    uint32_t chosen, t = 0;
    if (level == 0) {
      chosen = __hip_atomic_load(&a.fast_level[a.run_parity], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      t = atomicAdd(&a.tickets[TICKET0], 1u);
    } else {
      chosen = __hip_atomic_load(&a.fast_level[a.run_parity], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      if (chosen == level)
        t = atomicAdd(&a.tickets[4 * level + TICKET0], 1u);
    }
    F.misc[M_TICKET] = chosen == level ? t : 0xFFFFFFFFu;
  }
  __syncthreads();
  if (uni(F.misc[M_TICKET]) == 0xFFFFFFFFu)
    return; // this run's workgroups need another LDS level: that launch does the work
  // ticket -> block: the blocks of the launch's streams interleaved (each stream's in
  // order, so every predecessor holds an earlier ticket).  A workgroup waits for ALL its
  // stream's workgroups in flight; with one stream after the other that is everything on
  // the chip, and it pays the slowest of ~1000 -- interleaved, 1000 / streams.
  const uint4 bs = a.fast_order[uni(F.misc[M_TICKET])];
  const uint32_t b = uni(bs.x), s = uni(bs.y), fb_now = uni(bs.w);
  uint32_t tz_now = uni(bs.z);
#endif
  // The workgroup's image and the stream's flags are asked for NOW, next to the stream's
  // record: the head of a workgroup is a chain of dependent loads (ticket -> block ->
  // stream -> flags -> table and image, 0.7-1.5 us each); the ticket's entry names the
  // table, so nothing waits for the record.  (lj_load_image's loads for LF_BW rows, reversed: five uint4 a lane.)
  const uint4* __restrict__ img_src = a.unstuffed + size_t(b) * LJ_IMG_U4;
  const uint32_t flags_now = a.results[s].flags;
  // (Every load of the head UNCONDITIONAL, on an address that is valid for every lane, and
  // nothing done with a loaded value before the last of them is issued.  Until round 5 the
  // fifth image load, the guess and the K0 words sat under lane conditions; the compiler put
  // the first use of a loaded value -- a register copy, the guess's mask -- INTO those blocks,
  // each behind an s_waitcnt vmcnt(0): the head was four memory round trips one after the
  // other instead of one, 4-5 us of a workgroup's 25.)
  static_assert(5 * LJ_T <= LJ_IMG_U4, "five uint4 a lane stay inside the block's image");
  uint4 im[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#ifndef RSX_LF_PLAIN_MEM // (the un-stuffed image is read once: non-temporal, like the pixel stores --
                         // together 1.5-2 % on cfg 3, cfg 4 and the Nikon legs, profiles/r06/ab/nontemporal.txt)
    const lf_u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const lf_u32x4*>(img_src + k * LJ_T + j));
    im[k] = make_uint4(q[0], q[1], q[2], q[3]);
#else
    im[k] = img_src[k * LJ_T + j];
#endif
  }
  // (the fifth: row 16 for the first 64 lanes; the others ask for their first chunk again --
  // a line they hold -- instead of rows 17..19, which nobody reads: 47 MB of HBM reads on cfg 3)
#ifdef RSX_LF_FIFTH_PLAIN // (experiment: rows 17..19 read as well)
  im[4] = img_src[4 * LJ_T + j];
#else
  im[4] = img_src[j < LF_BW * LJ_T / 4 - 4 * LJ_T ? 4 * LJ_T + j : j];
#endif
  const uint32_t ob_now = reinterpret_cast<const uint32_t*>(img_src + (LJ_BW / 4) * LJ_T)[j];
  // (a stream's subsequences are numbered from first_block * LJ_OWN: the guesses too; lane 0
  // reads lane 1's and ignores it)
  const uint32_t guess_raw = a.sub_start[size_t(b) * LJ_OWN + uint32_t(j >= 1 ? j - 1 : 0)];
  // Symbol base (round 4): K0 has counted every workgroup's symbols under the very entry
  // states this kernel decodes from (its chain's fixed point), so the index of the
  // workgroup's first symbol is a SUM the workgroup reads when it starts -- the words of
  // its stream's workgroups in front of it, 16 per lane in flight next to the image -- and
  // not the end of a look-back over their decodes (3.6-3.9 us of waiting for the slowest of
  // the nearest predecessors, every one of them waiting the same way).  What K0 cannot
  // vouch for is marked in the word ("flagged": its own estimate of the workgroup's entry
  // state is not what the predecessor's chain arrives at -- 1.7 % --, a code outside the
  // 10-bit table, data that does not synchronise); those workgroups' true counts are asked
  // for below, and they alone.  lj_scan_kernel checks every base afterwards.
  const uint32_t lb_now = b - fb_now;
  u64 kw_mine = 0, kwv[LF_K0_IT];
  {
    const u64* kw = a.k0w + fb_now;
    kw_mine = kw[lb_now];
#pragma unroll
    for (int it = 0; it < LF_K0_IT; ++it) {
      // (words at and behind the workgroup's own are not its predecessors': dropped below)
      const uint32_t k = uint32_t(it) * uint32_t(LJ_T) + uint32_t(j);
      kwv[it] = kw[k < lb_now ? k : lb_now];
    }
  }
#ifndef RSX_LF_TICKETS
  if (uniform_plan) {
    const uint32_t tb = uni(tbv), tp = uni(tpv);
    tz_now = lf_table_word(tb, tp & 15u, (tp >> 8) & 15u, (tp >> 16) & 15u, (tp >> 24) & 15u);
  }
#endif
  const uint32_t table_base = tz_now & 0xFFFFu;
  uint4 lut_now[MT ? 4 : 2];
  {
    if constexpr (PT) {
      // (a table per phase: the 2-byte form of each phase's 10-bit LUT, 8 bytes a lane and table,
      // built on the host -- ljpeg_build_fast_table16; phases beyond N ask for phase 0's again)
      const uint2* t16 = reinterpret_cast<const uint2*>(a.fast_tabs16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t t = (tz_now >> (16 + 4 * (k < N ? k : 0))) & 15u;
        const uint2 v = t16[size_t(table_base + t) * 256 + uint32_t(j)];
        if (k < 2) {
          lut_now[0].x = k == 0 ? v.x : lut_now[0].x;
          lut_now[0].y = k == 0 ? v.y : lut_now[0].y;
          lut_now[0].z = k == 1 ? v.x : lut_now[0].z;
          lut_now[0].w = k == 1 ? v.y : lut_now[0].w;
        } else {
          lut_now[1].x = k == 2 ? v.x : lut_now[1].x;
          lut_now[1].y = k == 2 ? v.y : lut_now[1].y;
          lut_now[1].z = k == 3 ? v.x : lut_now[1].z;
          lut_now[1].w = k == 3 ? v.y : lut_now[1].w;
        }
      }
    } else {
      const uint32_t t_even = MT ? ((tz_now >> 16) & 15u) : 0u;
      const uint4* src =
          reinterpret_cast<const uint4*>(a.fast_tabs + size_t(table_base + t_even) * 1024);
      lut_now[0] = src[j];
      lut_now[1] = src[j + LJ_T];
      if constexpr (MT) {
        const uint4* srb = reinterpret_cast<const uint4*>(
            a.fast_tabs + size_t(table_base + ((tz_now >> 20) & 15u)) * 1024);
        lut_now[2] = srb[j];
        lut_now[3] = srb[j + LJ_T];
      }
    }
  }
  FastStream S = lf_stream(a.streams[s]);
  // (Pin: every load above is ISSUED before the first exit below.  Left alone the compiler
  // sinks the loads whose values the exits do not need -- tables, image -- behind them, i.e.
  // behind the wait for the stream's record: one more round trip in a row.)
  // (whole 16-byte registers: pinned by one component the compiler splits the load in two)
  auto v4 = [](const uint4& x) -> lf_u32x4 { return lf_u32x4{x.x, x.y, x.z, x.w}; };
  asm volatile("" ::"v"(v4(im[0])), "v"(v4(im[1])), "v"(v4(im[2])), "v"(v4(im[3])),
               "v"(v4(im[4])), "v"(ob_now), "v"(guess_raw), "v"(kw_mine), "v"(flags_now),
               "v"(v4(lut_now[0])), "v"(v4(lut_now[1])), "v"(v4(lut_now[MT ? 2 : 0])),
               "v"(v4(lut_now[MT ? 3 : 1])));
  asm volatile("" ::"v"(kwv[0]), "v"(kwv[1]), "v"(kwv[2]), "v"(kwv[3]), "v"(kwv[4]), "v"(kwv[5]),
               "v"(kwv[6]), "v"(kwv[7]), "v"(kwv[8]), "v"(kwv[9]), "v"(kwv[10]), "v"(kwv[11]),
               "v"(kwv[12]), "v"(kwv[13]), "v"(kwv[14]), "v"(kwv[15]));
  static_assert(LF_K0_IT == 16, "the pin lists the K0 words one by one");
#ifndef RSX_LF_TICKETS
  if (uni(chosen_now) != level)
    return; // this run's workgroups need another LDS level: that launch does the work
#endif
  if (int(S.fast_n) != N || int(S.tm) != TM || (S.diffs != 0u) != DIFF || (S.nk != 0u) != NK)
    return; // (workgroup-uniform)
  NkOut nko{};
  if constexpr (NK) {
    // (a row of the stream is a row of the image, all of it kept, from column 0)
    S.kind = 0;
    S.keep = S.RS;
    S.out_x = 0;
    S.n_strips = 0;
  }
  if constexpr (DIFF) {
    // (the differences are ONE run: a "row" as long as the stream, written from its first symbol)
    S.kind = 0;
    S.RS = 0xFFFFFFF0u;
    S.keep = 0xFFFFFFF0u;
    S.out_x = S.out_y = S.pitch = 0;
    S.n_strips = 0;
  }
  const uint32_t lb = b - S.first_block;
  // A stream that some workgroup has given up on (periodic data, an invalid code, ...) is
  // redone by the multi-kernel pipeline anyway: leave records the workgroups in flight
  // can walk over and go.
  if (uni(flags_now) & FL_SLOW) {
    if (j == 0) {
      u64* p = a.lb + size_t(b) * LF_LB_WORDS;
      for (int k = 0; k < LF_LB_WORDS; ++k)
        lb_store(p + k, LB_VALID);
      a.block_start[b] = 0;
      a.block_exit[b] = 0;
      a.block_sum[b] = 0;
      a.block_flags[b] = 0;
      a.block_psum[b] = make_uint2(0, 0);
    }
    return;
  }
#ifdef RSX_EXPERIMENT
  if (a.dbg && j == 0)
    a.dbg[size_t(b) * 16] = t_start;
#endif
  LF_STAMP(1);

  // tables + image
  if constexpr (MT) {
    // (entries 2j, 2j + 1 and 512 + 2j, 512 + 2j + 1 of either table, in their 4-byte form)
    auto e4 = [](uint32_t x, uint32_t y) -> uint32_t {
      return (x & 0x80000000u) ? LF_MT_SPECIAL : ((y << 16) | (x & 0x7FFu));
    };
    uint2* dst = reinterpret_cast<uint2*>(smem + LF_OFF_LUT);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint4 lo = lut_now[2 * t], hi = lut_now[2 * t + 1];
      dst[512 * t + j] = make_uint2(e4(lo.x, lo.y), e4(lo.z, lo.w));
      dst[512 * t + 256 + j] = make_uint2(e4(hi.x, hi.y), e4(hi.z, hi.w));
    }
  } else if constexpr (PT) {
    // (phase k's table at LDS address 2048 k: entries 4j .. 4j + 3 of each)
    uint2* dst = reinterpret_cast<uint2*>(smem + LF_OFF_LUT);
    dst[j] = make_uint2(lut_now[0].x, lut_now[0].y);
    dst[256 + j] = make_uint2(lut_now[0].z, lut_now[0].w);
    if (N >= 3)
      dst[512 + j] = make_uint2(lut_now[1].x, lut_now[1].y);
    if (N >= 4)
      dst[768 + j] = make_uint2(lut_now[1].z, lut_now[1].w);
  } else {
    uint4* dst = reinterpret_cast<uint4*>(smem + LF_OFF_LUT);
    dst[j] = lut_now[0];
    dst[j + LJ_T] = lut_now[1];
  }
  if (j == 0) {
    F.misc[M_SLOW] = 0;
    F.misc[M_NSIDE] = 0;
    F.misc[M_UNRES] = 0xFFFFu;
    F.misc[M_UNRESB] = 0;
  }
  {
    uint4* dst = reinterpret_cast<uint4*>(F.B);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = k * LJ_T + j; // uint4 i of the image = dwords 4 * (i % 64) .. of row i / 64
      if (i < LF_BW * LJ_T / 4)
        dst[(LF_BW - 1 - (i >> 6)) * (LJ_T / 4) + (i & 63)] = im[k];
    }
    F.ob[j] = uint16_t(ob_now);
  }
  // the CR2 strips (into LDS behind the bit delay: their region at the end of the allocation
  // is nobody else's; asked for in front of the rows, as they were, a global-memory round
  // trip of 1 us stood in every workgroup's way.  In LDS because a load behind the pixel
  // stores would wait for their acknowledgements.)
  static_assert((MAX_CR2_STRIPS + 1) * sizeof(Cr2Strip) / 4 <= 2 * LJ_T, "two dwords a lane");
  uint32_t strip_w0 = 0, strip_w1 = 0;
  const uint32_t strip_nw = S.kind == 1 ? (S.n_strips + 1) * uint32_t(sizeof(Cr2Strip) / 4) : 0u;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.strips + S.strip_base);
    if (uint32_t(j) < strip_nw)
      strip_w0 = src[j];
    if (uint32_t(j) + uint32_t(LJ_T) < strip_nw)
      strip_w1 = src[j + LJ_T];
  }
  // the symbols in front of the workgroup as K0 counted them, and which of those
  // workgroups are flagged (own estimate ^ true entry = the "on record" bit and nothing else
  // -- "uncertain" sits in the own half -- means neither)
  uint32_t kacc = 0, kflag = 0;
#pragma unroll
  for (int it = 0; it < LF_K0_IT; ++it) {
    const u64 w = uint32_t(it) * uint32_t(LJ_T) + uint32_t(j) < lb_now ? kwv[it] : 0ull;
    kacc += uint32_t(w);
    const uint32_t hi = uint32_t(w >> 32);
    kflag |= (((hi >> 16) ^ hi) & 0xFFFFu) != 0x8000u ? (1u << it) : 0u;
  }
  {
    const uint32_t nval =
        lb_now > uint32_t(j) ? min(uint32_t(LF_K0_IT), (lb_now - uint32_t(j) + 255u) >> 8) : 0u;
    kflag &= (1u << nval) - 1u;
  }
  lds_barrier();
  LF_STAMP(2);
  // delay the lane's column by one bit (see the header): dword k := d[k-1] : d[k] >> 1
  {
    uint32_t prev = 0;
#pragma unroll
    for (int k = 0; k < LF_BW; ++k) {
      uint32_t* p = &F.B[(LF_BW - 1 - k) * LJ_T + j];
      const uint32_t d = *p;
      *p = __builtin_amdgcn_alignbit(prev, d, 1);
      prev = d;
    }
  }
  // (streams of more than LF_K0_IT * 256 workgroups -- 67 MB of entropy-coded data --: the
  // rest of the words, group by group)
  for (uint32_t k0 = uint32_t(LF_K0_IT) * uint32_t(LJ_T); k0 < lb_now; k0 += uint32_t(LJ_T)) {
    const uint32_t k = k0 + uint32_t(j);
    if (k < lb_now) {
      const u64 w = a.k0w[fb_now + k];
      kacc += uint32_t(w);
      if (lf_k0_flagged(w)) {
        u64 g = 0;
        for (uint32_t spins = 0; spins < LF_SPIN_LIMIT_K0 && !(g & LB_VALID); ++spins)
          g = lb_load(a.lb + size_t(fb_now + k) * LF_LB_WORDS);
        if (g & LB_VALID)
          kacc += uint32_t(g);
        else
          F.misc[M_SLOW] = 6;
      }
    }
  }
  if (uint32_t(j) < strip_nw)
    reinterpret_cast<uint32_t*>(F.strips)[j] = strip_w0;
  if (uint32_t(j) + uint32_t(LJ_T) < strip_nw)
    reinterpret_cast<uint32_t*>(F.strips)[j + LJ_T] = strip_w1;
  lds_barrier();
  LF_STAMP(3);
  const uint32_t own_bits = F.ob[j];
  // LDS address of the row of dword 0 of a column
  const uint32_t vbase_own = lds_addr(&F.B[(LF_BW - 1) * LJ_T + j]);

  // 1. the start guess: left by lj_unstuff_kernel (a parse of the three slots before the
  // lane's from bit 0; made in this kernel, from one or two slots, every slot parsed added
  // 6 us to the workgroup's lifetime)
  const uint32_t guess = j >= 1 ? guess_raw : 0u;
  LF_STAMP(4);
  uint32_t start = guess & SMASK;
  // slot 1 starts where the predecessor workgroup's chain ends (its lane 255 left it in this
  // workgroup's word), the stream's first slot where the stream does
  const uint32_t e1_word = uni(uint32_t(kw_mine >> 48));
  const uint32_t k0_cnt_mine = uni(uint32_t(kw_mine));
  if (j == 1 && (e1_word & 0x8000u))
    start = e1_word & SMASK;
  if (j == 1 && lb == 0)
    start = S.start_bit;
  // (Tried for two tables, where K0's estimate of the entry state was off in 12 % of the
  // workgroups before its hand-over: the workgroup that sees the mismatch follows the true
  // chain and K0's through its first slots itself -- two lanes, lengths only, 3.5 us a slot --,
  // puts the difference of their counts out at once and starts the slots in between from the
  // true states.  It took the re-decode rounds away and the wait for a predecessor's count from
  // 17 to 5 us, and moved it into look-back 1: a workgroup that is 3.5 us late there holds up
  // everyone behind it in flight just the same.  K0's hand-over leaves 1.4 % mis-estimated,
  // as with one table, and they are left to the rounds.)
  // 2. decode, keeping the running sums
  uint32_t R[LF_NR];
  FastState fs;
  fs.Pn = uint32_t(-32) - 32u * (start & ST_OFF_MASK);
  fs.n = 0;
  fs.acc[0] = fs.acc[1] = fs.acc[2] = fs.acc[3] = 0;
  fs.ev = 0;
  fs.spec = 0;
  const uint32_t pend = uint32_t(-32) - 32u * own_bits;
  if (j == 0)
    fs.Pn = pend; // (slot 0 belongs to the previous workgroup: nothing to decode)
  if (LF_ABLATE & 16u)
    fs.Pn = pend;
  // (per-phase tables: the phase of the lane's first symbol, 0 .. N - 1)
  const uint32_t ph0 = PT ? ((start >> ST_PHASE_SHIFT) & 3u) : 0u;
  if constexpr (MT) {
    // (the LUT of the lane's even-numbered symbols, of its odd-numbered ones)
    const uint32_t lut0 = (start & 64u) ? 4096u : 0u;
    lf_groups<N, 0, 1>(fs, vbase_own, pend, R, lut0, lut0 ^ 4096u);
  } else if constexpr (PT) {
    // (the LUT of the lane's symbols k mod N == 0, 1, 2, 3: phase (ph0 + k) mod N, 2 KB each)
    uint32_t lb_[4];
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
      lb_[k] = lf_pmod(ph0 + k, S.tp) << 11;
    lf_groups<N, 0, 2>(fs, vbase_own, pend, R, lb_[0], lb_[1], lb_[2], lb_[3]);
  } else {
    lf_groups<N, 0, 0, DIFF>(fs, vbase_own, pend, R);
  }
  LF_STAMP(5);
  // the granules of the flagged workgroups in front (1.7 % of them: 0.24 per lane on
  // average, the first two of a lane asked for now): in flight behind the rounds and scans
  u64 kg0 = 0, kg1 = 0;
  uint32_t kit0 = 0, kit1 = 0;
  if (kflag != 0u) {
    kit0 = uint32_t(__builtin_ctz(kflag));
    kg0 = lb_load(a.lb + size_t(fb_now + kit0 * uint32_t(LJ_T) + uint32_t(j)) * LF_LB_WORDS);
    const uint32_t rest = kflag & (kflag - 1u);
    if (rest != 0u) {
      kit1 = uint32_t(__builtin_ctz(rest));
      kg1 = lb_load(a.lb + size_t(fb_now + kit1 * uint32_t(LJ_T) + uint32_t(j)) * LF_LB_WORDS);
    }
  }
  bool need_redo = false;
  {
    // (two tables: a special entry sends the position far past the slot's end, lf_step_mt)
    // (per-phase tables: a special entry has bit 15, and the lane has read one if the OR of
    // its entries has)
    const bool special = MT ? (fs.Pn <= pend && ((pend - fs.Pn) >> 5) >= 64u)
                            : (PT ? (fs.spec & 0x8000u) != 0u : !(fs.Pn & 0x80000000u));
    const bool over = !special && fs.Pn > pend; // more than LF_MAXSYM symbols
    uint32_t ex = special ? ST_ERR : ((pend - fs.Pn) >> 5);
    if (MT && !special) // the table the NEXT symbol takes
      ex |= (((start >> ST_PHASE_SHIFT) ^ fs.n) & 1u) << ST_PHASE_SHIFT;
    if (PT && !special) // ... its phase
      ex |= lf_pmod(ph0 + fs.n, S.tp) << ST_PHASE_SHIFT;
    if (over) {
      ex = ST_ERR;
      atomicMin(&F.misc[M_UNRES], uint32_t(j));
    }
    need_redo = special;
    if (j >= 1)
      F.rec[j] = rec_make(start, ex, fs.n);
    F.sm[j] = make_uint2(pack16(fs.acc[0], fs.acc[1]), pack16(fs.acc[2], fs.acc[3]));
    // slot 0 stands for the workgroup's entry state: its "exit" is what lane 1 assumed
    // (written by lane 1 itself, in front of the barrier the first round starts from)
    if (j == 1)
      F.rec[0] = rec_make(0, start, 0);
    if (j == 0)
      F.misc[M_LIST] = 0;
  }
  lds_barrier();

  const uint32_t gsub = S.first_subseq + lb * LJ_OWN + uint32_t(j - 1); // j >= 1
  uint32_t my_cnt = 0, before = 0, cnt_wg = 0, base = 0;
  uint2 my_sums = make_uint2(0, 0), pex = make_uint2(0, 0), S_wg = make_uint2(0, 0);
  uint32_t published_exit = 0;
  // Subsequences that are re-decoded get a side-buffer entry of their own for the rest
  // of the kernel and their owners fetch the sums ONCE, after all rounds: a fetch inside
  // the loops makes every register of R loop-carried, and the compiler then keeps two
  // copies of the 64 (measured: 180 VGPRs).
  int my_entry = -1;
  {
    constexpr int attempt = 0; // (rounds 1-3 repaired a wrong entry state in a second attempt, after
                               // look-back 0 had told the workgroup; see "symbol base")
    // 3. Jacobi rounds with a dense list (lj_sync_kernel's scheme)
    uint32_t rounds = 0;
    while (true) {
      // (the first round of the first attempt starts from the barrier behind the decode:
      // records, slot 0's entry and the empty list are all in front of it)
      const bool first_pass = attempt == 0 && rounds == 0;
      if (!first_pass) {
        if (j == 0)
          F.misc[M_LIST] = 0;
        lds_barrier();
      }
      const uint32_t my_su = rec_su(F.rec[j]);
      const uint32_t want = j >= 1 ? rec_st(F.rec[j - 1]) : my_su;
      const bool chained = own_bits != 0u && j >= 1;
      // After the first round only the HEAD of a run of inconsistent slots is redone.
      // Plain Jacobi re-decodes slot j from the exit its predecessor had BEFORE this round:
      // where the data does not synchronise (a constant region, its zero-difference code
      // over and over: a slot started off the symbol grid can leave off the grid) that
      // stale exit breaks slot j in the round that repairs slot j - 1, and the error travels
      // down the region one slot per round, the repair one slot behind it.
      bool pred_stale = false;
      if (rounds != 0 && j >= 2) {
        const uint32_t pp = rec_st(F.rec[j - 2]);
        pred_stale = rec_su(F.rec[j - 1]) != pp && !(pp & ST_ERR) && F.ob[j - 1] != 0;
      }
      const bool listed =
          chained && (need_redo || (want != my_su && !(want & ST_ERR) && !pred_stale));
      // side-buffer entries in slot order: what follows the last delivered symbol (zeros
      // behind the end-of-image marker, trailing bytes: periodic, every slot inconsistent)
      // must not take them from the slots in front of it
      {
        const bool ask = listed && my_entry < 0;
        const unsigned long long am = __ballot(ask);
        if (lane == 0)
          F.misc[M_WNE + wv] = uint32_t(__builtin_popcountll(am));
        const uint32_t handed = F.misc[M_NSIDE];
        lds_barrier();
        uint32_t first = handed, all = handed;
        for (int w = 0; w < 4; ++w) {
          const uint32_t t = F.misc[M_WNE + w];
          first += w < wv ? t : 0u;
          all += t;
        }
        // (the first round leaves four entries for slots that turn up later, further in front)
        const uint32_t limit = rounds == 0 ? uint32_t(LF_NSIDE) - 4u : uint32_t(LF_NSIDE);
        if (ask) {
          const uint32_t k = first + uint32_t(__builtin_popcountll(am & ((1ull << lane) - 1ull)));
          my_entry = k < limit ? int(k) : -1;
        }
        if (j == 0)
          F.misc[M_NSIDE] = all < limit ? all : (handed > limit ? handed : limit);
        // Nothing to re-decode anywhere (99 % of the workgroups): in the first pass nobody
        // holds an entry yet, so "nobody asks" (the four counts every lane has just read)
        // is "nobody is listed" -- no list, no third barrier.  (The same test with
        // __syncthreads_or brings static LDS along: see the layout's note.)
        if (first_pass && all == handed) {
          need_redo = false;
          break;
        }
      }
      if (listed && my_entry >= 0)
        F.list[atomicAdd(&F.misc[M_LIST], 1u)] = uint16_t(j | (my_entry << 8));
      need_redo = false;
      lds_barrier();
      const uint32_t nl = (LF_ABLATE & 64u) ? 0u : uni(F.misc[M_LIST]);
      // (more re-decodes than the side buffer has entries, or data that does not
      // synchronise: what is left inconsistent is dealt with below)
      if (nl == 0 || ++rounds > LF_MAX_ROUNDS)
        break;
#ifdef RSX_EXPERIMENT
      if (j == 0) {
        atomicAdd(&a.results[s].stat_rounds, 1u);
        atomicAdd(&a.results[s].stat_redo, nl);
      }
#endif
      uint32_t idx = 1, w = 0, e = 0, c = 0;
      uint2 sums = make_uint2(0, 0);
      bool ovf = false;
      const bool mine = wv == 0 && uint32_t(lane) < nl; // (nl <= LF_NSIDE <= 64)
      if (wv == 0) {
        const uint32_t le = mine ? uint32_t(F.list[lane]) : 1u;
        idx = le & 0xFFu;
        w = rec_st(F.rec[idx - 1]);
        if (w & ST_ERR) // (listed for its own sake: from the state it started from)
          w = rec_su(F.rec[idx]) & SMASK;
        lf_redecode<N, TM, DIFF>(F, a.tables + S.table_base, S.tabsel, S.tp, int(idx), w, F.ob[idx],
                           lds_addr(F.side) + (le >> 8) * LF_SIDE_STRIDE, mine, e, c, sums, ovf);
      }
      lds_barrier(); // every read of the records precedes the updates
      if (mine) {
        F.rec[idx] = rec_make(w, e, c > 0xFFFu ? 0xFFFu : c);
        F.sm[idx] = sums;
        if (ovf)
          atomicMin(&F.misc[M_UNRES], idx);
      }
    }
    // Slots the rounds have left inconsistent (periodic data, more re-decodes than side
    // entries, more than LF_MAXSYM symbols): everything from the first of them on is
    // unknown.  That only matters if the stream's delivered symbols reach that far --
    // what follows the last of them (the zeros the bit pump feeds after the end-of-image
    // marker, any trailing bytes of the input) is periodic as a rule and nobody's business.
    {
      const uint32_t su = rec_su(F.rec[j]);
      const uint32_t wn = j >= 1 ? rec_st(F.rec[j - 1]) : su;
      if (own_bits != 0u && j >= 1 && wn != su && !(wn & ST_ERR))
        atomicMin(&F.misc[M_UNRES], uint32_t(j));
    }

    // per slot: symbols before it, running sums before it (workgroup-relative phases)
    const uint32_t my_rec = F.rec[j];
    my_cnt = j >= 1 ? rec_cn(my_rec) : 0u;
    if (my_cnt > uint32_t(LF_MAXSYM))
      my_cnt = uint32_t(LF_MAXSYM); // (flagged slow already)
    my_sums = j >= 1 ? F.sm[j] : make_uint2(0u, 0u);
    if (N == 1)
      my_sums = make_uint2(my_sums.x & 0xFFFFu, 0u);
    if (N == 2)
      my_sums.y = 0u;
    if (N == 3)
      my_sums.y &= 0xFFFFu;
    // (the symbol base rides on the same barrier: K0's counts of the workgroups in front, read
    // at the start, + the corrections of the flagged ones asked for behind the decode.  What
    // is still missing then -- flagged workgroups that were in flight themselves -- is asked
    // for BEHIND the barrier, after this workgroup's own correction is out: asked for in
    // front of it, as rounds 4's first version did, a flagged workgroup made its successors
    // wait for ITS predecessors -- nothing with 1.7 % of them flagged, a chain through the
    // whole stream with two tables, where one in five is: 26 us of a workgroup's 49.)
    {
      if (kg0 & LB_VALID) {
        kacc += uint32_t(kg0);
        kflag &= ~(1u << kit0);
      }
      if (kg1 & LB_VALID) {
        kacc += uint32_t(kg1);
        kflag &= ~(1u << kit1);
      }
      // (one table: 1.5 % of the workgroups are flagged and hardly any of them is slow -- asking
      // right here, in front of the barrier, is 5 % of the kernel faster on cfg 4 than the
      // second barrier the other order needs now and then)
      if constexpr (!MT) {
        uint32_t spins = 0;
        while (__any(kflag != 0u)) {
          if (kflag != 0u) {
            const uint32_t it = uint32_t(__builtin_ctz(kflag));
            const u64 g = lb_load(a.lb + size_t(fb_now + it * uint32_t(LJ_T) + uint32_t(j)) * LF_LB_WORDS);
            if (g & LB_VALID) {
              kacc += uint32_t(g);
              kflag &= kflag - 1u;
            }
          }
          if (++spins > LF_SPIN_LIMIT_K0) {
            F.misc[M_SLOW] = 6;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        kflag = 0u;
      }
      const bool wave_pending = MT && __any(kflag != 0u);
      const uint32_t part = wave_sum_u32(kacc);
      if (lane == 0) {
        F.misc[M_LBX + wv] = part;
        F.misc[M_LBX + 4 + wv] = wave_pending ? 1u : 0u;
      }
    }
    // (both scans in front of ONE barrier: the sums are rotated by the symbols before the lane
    // inside its WAVEFRONT first -- rotations add up --, the symbols of the wavefronts in
    // front are applied to the scanned values afterwards)
    const uint32_t incl = wave_scan_u32(my_cnt, lane);
    const uint32_t lbef = incl - my_cnt;
    const uint2 r_l = lj_rot_fields<N>(my_sums, lf_mod<N>(lbef));
    const uint2 pincl_l = wave_scan_pk2(r_l, lane);
    if (lane == 63) {
      F.misc[M_WCNT + wv] = incl;
      F.misc[M_WSUM + 2 * wv] = pincl_l.x;
      F.misc[M_WSUM + 2 * wv + 1] = pincl_l.y;
    }
    lds_barrier();
    uint32_t wprev = 0, wrun = 0;
    uint2 sprev = make_uint2(0, 0);
    S_wg = make_uint2(0, 0);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t tc = uni(F.misc[M_WCNT + w]);
      const uint2 ts = lj_rot_fields<N>(
          make_uint2(uni(F.misc[M_WSUM + 2 * w]), uni(F.misc[M_WSUM + 2 * w + 1])),
          lf_mod<N>(wrun));
      if (w < wv) {
        wprev += tc;
        sprev = pk_add2(sprev, ts);
      }
      S_wg = pk_add2(S_wg, ts);
      wrun += tc;
    }
    before = lbef + wprev;
    pex = pk_add2(lj_rot_fields<N>(pk_sub2(pincl_l, r_l), lf_mod<N>(wprev)), sprev);
    if (uint32_t(j) == F.misc[M_UNRES])
      F.misc[M_UNRESB] = before;
    cnt_wg = wrun;
    const uint32_t exit_now = uni(rec_st(F.rec[LJ_T - 1]));
    // 4a. the workgroup's granule as soon as its symbols are known: what it decoded beyond
    // (or short of) K0's count.  Only successors that find this workgroup FLAGGED in K0's
    // words read it.
    if (j == 0)
      lb_store(a.lb + size_t(b) * LF_LB_WORDS, LB_VALID | u64(uint32_t(cnt_wg - k0_cnt_mine)));

    // 4. the symbol base: K0's counts of the workgroups in front (read at the start) + the
    // corrections of the flagged ones.  Those still in flight when this workgroup started --
    // flagged ones among its ~128 nearest predecessors: two on average, and only the very
    // nearest can still be decoding -- are asked again now.
    published_exit = exit_now;
    LF_STAMP(6);
    if (MT && uni(F.misc[M_LBX + 4] | F.misc[M_LBX + 5] | F.misc[M_LBX + 6] | F.misc[M_LBX + 7])) {
      uint32_t spins = 0;
      while (__any(kflag != 0u)) {
        if (kflag != 0u) {
          const uint32_t it = uint32_t(__builtin_ctz(kflag));
          const u64 g = lb_load(a.lb + size_t(fb_now + it * uint32_t(LJ_T) + uint32_t(j)) * LF_LB_WORDS);
          if (g & LB_VALID) {
            kacc += uint32_t(g);
            kflag &= kflag - 1u;
          }
        }
        if (++spins > LF_SPIN_LIMIT_K0) {
          F.misc[M_SLOW] = 6;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      const uint32_t part = wave_sum_u32(kacc);
      if (lane == 0)
        F.misc[M_LBX + 8 + wv] = part;
      lds_barrier();
      base = uni(F.misc[M_LBX + 8] + F.misc[M_LBX + 9] + F.misc[M_LBX + 10] + F.misc[M_LBX + 11]);
    } else {
      base = uni(F.misc[M_LBX] + F.misc[M_LBX + 1] + F.misc[M_LBX + 2] + F.misc[M_LBX + 3]);
    }
    if (LF_ABLATE & 4u)
      base = lb * 15500u;
    LF_STAMP(7);
  }
  if (__any(my_entry >= 0 && my_entry < LF_NSIDE)) {
    if (my_entry >= 0 && my_entry < LF_NSIDE) {
      // (dword by dword: a uint4 view turns R into 16 vectors for the compiler)
      const uint32_t sa = lds_addr(F.side) + uint32_t(my_entry) * LF_SIDE_STRIDE;
#pragma unroll
      for (int q = 0; q < LF_NR; ++q)
        R[q] = *(lds_u32p)(sa + 4u * q);
    }
  }
  // (the records the per-stream bookkeeping kernels read are stored at the very end: on
  // gfx9 a store in front of the look-back's loads makes their s_waitcnt wait for its
  // acknowledgement, too)
  const uint32_t my_rec_final = F.rec[j];
  const uint32_t entry_final = uni(rec_st(F.rec[0]));
  const uint64_t needed = S.needed;
  const uint32_t i0 = base + before; // index of the lane's first symbol
  uint32_t cnt_eff = my_cnt;
  if (uint64_t(i0) >= needed)
    cnt_eff = 0;
  else if (uint64_t(i0) + cnt_eff > needed)
    cnt_eff = uint32_t(needed - i0);
  {
    const uint32_t my_start = rec_su(my_rec_final), my_exit = rec_st(my_rec_final);
    // an invalid code inside the delivered range: the slow path reports it the
    // reference's way (PrefixCodeLookupDecoder.h:152-155)
    if (j >= 1 && (my_exit & ST_ERR) && uint64_t(i0) + my_cnt < needed && own_bits != 0)
      F.misc[M_SLOW] = 7;
    // lj_consumed_kernel needs the bit position at which the reference's last symbol
    // starts: exactly one lane of the stream owns it and walks there again
    if (needed >= 1 && needed - 1 >= i0 && needed - 1 < uint64_t(i0) + my_cnt && j >= 1) {
      const uint32_t target = uint32_t(needed - 1 - i0);
      uint32_t p2 = my_start & ST_OFF_MASK;
      uint32_t odd = TM != 0 ? ((my_start >> ST_PHASE_SHIFT) & (MT ? 1u : 3u)) : 0u;
      for (uint32_t t = 0; t < target; ++t) {
        const uint32_t w = lj_window<LF_BW>(F.B, j, p2 + 1u);
        p2 += lf_slow_entry(w, a.tables[S.table_base + ((S.tabsel >> (4u * odd)) & 15u)]) >> 10;
        if (MT)
          odd ^= 1u;
        if (PT)
          odd = odd + 1u == S.tp ? 0u : odd + 1u;
      }
      a.results[s].last_slot = lb * LJ_OWN + uint32_t(j - 1);
      a.results[s].last_pos = p2;
    }
  }
  LF_STAMP(8);

  // 5. geometry of the delivered symbols [base, lim) and the stream rows they touch
  const uint32_t RS = S.RS;
  uint64_t lim64 = uint64_t(base) + cnt_wg;
  if (lim64 > needed)
    lim64 = needed;
  const uint32_t lim = uint32_t(lim64);
  const bool any_out = lim > base;
  const uint32_t r0 = base / RS;
  const uint32_t r_end = any_out ? (lim - 1) / RS : r0;
  const uint32_t nr = r_end - r0 + 1;
  // (more rows than lanes, or more samples than the staging region holds: slow path)
  // (a workgroup past the last delivered symbol -- the padding rows of an overhanging DNG
  // tile -- delivers nothing)
  const bool fits = !any_out || (nr <= uint32_t(LF_RMAX) && (lim - base) <= F.stage_cap);
  if (!fits && j == 0)
    F.misc[M_SLOW] = 8;
  const uint32_t sb = LF_STAGE_BASE;
  lds_barrier(); // every lane is done with the image, the records and the side buffer
  LF_STAMP(9);
  // does a delivered symbol lie in the part of the workgroup the rounds did not finish?
  // (behind the barrier: M_UNRESB comes from the lane that owns slot M_UNRES)
  if (j == 0 && F.misc[M_UNRES] != 0xFFFFu && uint64_t(base) + F.misc[M_UNRESB] < needed) {
    F.misc[M_SLOW] = 3;
#ifdef RSX_EXPERIMENT
    a.results[s].pad3[0] = lb;
    a.results[s].pad3[1] = F.misc[M_UNRES] | (F.misc[M_UNRESB] << 16);
    a.results[s].pad3[2] = base;
#endif
  }

  // 6. staging: running sums + P before the lane = Ploc, in stream order.  (Tried in round
  // 4: in two phases -- first the few lanes that hold the first MCUs of the rows starting
  // here, then the LOCAL record of look-back 1, then everybody else behind it.  No gain: the
  // compiler's staging took 2 us whoever staged; the hand-written block of lf_stage takes
  // 0.4 us for everybody.)
  uint2 Cloc = make_uint2(0, 0), Vsum = make_uint2(0, 0);
  uint32_t lb1_flags = 0;
  uint2 lb1_al = make_uint2(0, 0);
  const uint2 S_abs = lj_rot_fields<N>(S_wg, lf_mod<N>(base));
  if (fits && !(LF_ABLATE & 1u)) {
    const uint2 pexrel =
        lj_rot_fields<N>(pex, lf_mod<N>(uint32_t(N) - lf_mod<N>(before)));
    const uint32_t ad = sb + 2u * before;
    // pairs that are written from the registers (a count clipped by `needed` writes one
    // sample more: nothing after it is delivered)
    const uint32_t nq = cnt_eff == my_cnt ? (cnt_eff >> 1) : ((cnt_eff + 1) >> 1);
    const uint32_t nqmax = wave_max_u32(nq);
    // (register pair q holds the lane's symbols 2q, 2q + 1, i.e. relative components 2q mod N,
    // 2q + 1 mod N: N = 1, 2 one constant, N = 4 two in turn, N = 3 three in turn -- (0, 1),
    // (2, 0), (1, 2))
    const uint32_t k0 = DIFF ? 0u : (N == 1 ? (pexrel.x & 0xFFFFu) * 0x10001u : pexrel.x);
    const uint32_t k1 = N == 4 ? pexrel.y
                               : (N == 3 ? pack16(pexrel.y, pexrel.x) : k0);
    const uint32_t k2 = N == 3 ? pack16(pexrel.x >> 16, pexrel.y) : k0;
    lf_stage<N, 0>(R, ad, nq, nqmax, k0, k1, k2);
    if ((cnt_eff & 1u) && cnt_eff == my_cnt) {
      const uint32_t k = cnt_eff - 1;
      const uint32_t v = fld(my_sums, lf_mod<N>(k)) + (DIFF ? 0u : fld(pexrel, lf_mod<N>(k)));
      *(lds_u16w)(ad + 2u * k) = uint16_t(v);
    }
  }
  lds_barrier();
  LF_STAMP(10);

  WalkStart walk0{0, 0, 0, 1, 0, 0, 0xFFFFFFFFu};
  if constexpr (DIFF) {
    // (differences: no rows, no predictor state to carry -- the constants of the one run are zero)
    if (j == 0)
      F.ctab[0] = make_uint2(0u, 0u);
    (void)Cloc;
    (void)Vsum;
    (void)lb1_flags;
    (void)lb1_al;
    (void)S_abs;
  } else {
  // 7. rows.  Lane t takes stream row r0 + t: for each component whose first-MCU symbol
  // r * RS + c lies in the workgroup, E = Ploc before it (the staged sample N back, 0 at
  // the workgroup's start) and D = its difference.
  {
    uint32_t ev[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
    if (fits && uint32_t(j) < nr && !(LF_ABLATE & 128u)) {
      const uint64_t rs = uint64_t(r0 + uint32_t(j)) * RS;
#pragma unroll
      for (uint32_t c = 0; c < uint32_t(N); ++c) {
        const uint64_t i = rs + c;
        if (i >= base && i < lim) {
          const uint32_t e = uint32_t(i - base);
          const uint32_t fv = *(const __attribute__((address_space(3))) uint16_t*)(sb + 2u * e);
          const uint32_t pv =
              e >= uint32_t(N)
                  ? uint32_t(*(const __attribute__((address_space(3))) uint16_t*)(sb + 2u * (e - N)))
                  : 0u;
          ev[c] = pv;
          dv[c] = (fv - pv) & 0xFFFFu;
        }
      }
    }
    const uint2 E = make_uint2(ev[0] | (ev[1] << 16), ev[2] | (ev[3] << 16));
    uint2 D = make_uint2(dv[0] | (dv[1] << 16), dv[2] | (dv[3] << 16));
    // (Nikon-type: the row's pair goes to the half of its parity)
    const bool rodd = NK && ((r0 + uint32_t(j)) & 1u) != 0u;
    if (rodd)
      D = make_uint2(0u, D.x);
    const uint2 dincl = wave_scan_pk2(D, lane);
    if (lane == 63) {
      F.misc[M_RSUM + 2 * wv] = dincl.x;
      F.misc[M_RSUM + 2 * wv + 1] = dincl.y;
    }
    lds_barrier();
    uint2 vex = pk_sub2(dincl, D);
    for (int w = 0; w < 4; ++w) {
      const uint2 t = make_uint2(uni(F.misc[M_RSUM + 2 * w]), uni(F.misc[M_RSUM + 2 * w + 1]));
      if (w < wv)
        vex = pk_add2(vex, t);
      Vsum = pk_add2(Vsum, t);
    }
    Cloc = NK ? make_uint2(pk_sub(rodd ? vex.y : vex.x, E.x), 0u) : pk_sub2(vex, E);
    if (uint32_t(j) < nr && nr <= uint32_t(LF_RMAX))
      F.ctab[j] = Cloc;
  }
  lds_barrier();
  LF_STAMP(11);
  // 8a. the LOCAL record of look-back 1 (S_abs: the sums of the workgroup's differences by
  // absolute component)
  {
    // which components start a row here, and from which table row their last start is
    uint32_t flags = 0;
    uint2 al = S_abs;
    if (fits && any_out) {
      uint32_t av[4] = {0, 0, 0, 0};
#pragma unroll
      for (uint32_t c = 0; c < uint32_t(N); ++c) {
        int tl = -1;
        const uint64_t il = uint64_t(r_end) * RS + c;
        if (il >= base && il < lim)
          tl = int(nr) - 1;
        else if (nr >= 2 && uint64_t(r_end - 1) * RS + c >= base)
          tl = int(nr) - 2;
        if constexpr (NK) {
          // (both halves carry the component's sum; the half of the started row's parity is set)
          av[c] = av[c + 2u] = fld(S_abs, c);
          if (tl >= 0) {
            const uint32_t h = 2u * ((r0 + uint32_t(tl)) & 1u);
            flags |= 1u << (c + h);
            av[c + h] = (fld(F.ctab[tl], c) + fld(S_abs, c)) & 0xFFFFu;
          }
        } else if (tl >= 0) {
          flags |= 1u << c;
          av[c] = (fld(F.ctab[tl], c) + fld(S_abs, c)) & 0xFFFFu;
        } else {
          av[c] = fld(S_abs, c);
        }
      }
      al = make_uint2(av[0] | (av[1] << 16), av[2] | (av[3] << 16));
    } else if constexpr (NK) {
      al = make_uint2(S_abs.x, S_abs.x);
    }
    constexpr int NW = NK ? 2 : (N + 1) / 2;
    if (j == 0) {
      u64* p = a.lb + size_t(b) * LF_LB_WORDS;
      lb_store(p + lb_wv(0), LB_VALID | Vsum.x);
      if (NW == 2)
        lb_store(p + lb_wv(1), LB_VALID | Vsum.y);
      if (NW == 2)
        lb_store(p + lb_wa(1), LB_VALID | (u64(flags) << 32) | al.y);
      lb_store(p + lb_wa(0), LB_VALID | (u64(flags) << 32) | al.x);
    }
    lb1_flags = flags;
    lb1_al = al;
  }
  if constexpr (NK) {
    // what the copy-out needs of the stream's Nikon record, and the dither generator's seed -- the
    // 24 bits at the stream's start --: asked for here, in the shadow of look-back 1's wait
    const NkStreamDev& K = a.nk[s];
    nko.dither = uni(K.uncorrected == 0u && K.pentax == 0u ? 1u : 0u);
    nko.limit_shift = uni(K.pentax != 0u && K.pentax < 15u ? K.pentax : 15u);
    nko.tab = a.nk_tables + uni(K.table_off);
    nko.rowpow = a.nk_rowpow + uni(K.rowpow_off);
    nko.colpow = a.nk_rowpow + uni(K.colpow_off);
    nko.flags = &a.results[s].flags;
    if (nko.dither) {
      const uint8_t* in0 = a.in_base + uni64(K.seed_offset);
      nko.seed = (uint32_t(in0[0]) << 16) | (uint32_t(in0[1]) << 8) | in0[2];
    }
  }
  const uint2 init = S.init;
  walk0 = lf_walk_start(F, S, base); // (the strips are in LDS since phase one)
  // 8. look-back 1
  uint2 T_in = init, V_in = init;
  {
    constexpr int NW = NK ? 2 : (N + 1) / 2;
    const uint32_t flags = lb1_flags;
    const uint2 al = lb1_al;
    bool ok = true;
    if (lb != 0 && !(LF_ABLATE & 8u)) {
      ok = lb1_walk<NK ? 4 : N, LF_LB1_WAVES>(a, F, b, S.first_block, init, j, &T_in, &V_in);
    }
    if (j == 0) {
      if (!ok)
        F.misc[M_SLOW] = 9;
      // the inclusive state
      const uint2 fm = fld_mask(flags);
      const uint2 T_out = pk_add2(al, sel2(fm, V_in, T_in));
      const uint2 V_out = pk_add2(V_in, Vsum);
      u64* p = a.lb + size_t(b) * LF_LB_WORDS;
      lb_store(p + lb_wc(0), LB_VALID | V_out.x);
      if (NW == 2)
        lb_store(p + lb_wc(1), LB_VALID | V_out.y);
      if (NW == 2)
        lb_store(p + lb_wt(1), LB_VALID | T_out.y);
      lb_store(p + lb_wt(0), LB_VALID | T_out.x);
    }
  }
  LF_STAMP(12);
  {
    // C(r, c): Vc_in + Cloc for the components whose first-MCU symbol is in this
    // workgroup, T_in for the row that is open when it starts
    if (fits && uint32_t(j) < nr) {
      uint2 C = pk_add2(V_in, Cloc);
      if constexpr (NK) // (the row's parity picks the half; row r0's open state likewise)
        C = make_uint2(pk_add(((r0 + uint32_t(j)) & 1u) ? V_in.y : V_in.x, Cloc.x), 0u);
      const uint2 T_row = NK && (r0 & 1u) ? make_uint2(T_in.y, 0u) : T_in;
      if (j == 0) {
        uint32_t present = 0;
#pragma unroll
        for (uint32_t c = 0; c < uint32_t(N); ++c)
          if (uint64_t(r0) * RS + c >= base)
            present |= 1u << c;
        C = sel2(fld_mask(present), C, T_row);
      }
      F.ctab[j] = C;
    }
  }
  }
  if (F.misc[M_SLOW] != 0 && j == 0) {
    atomicOr(&a.results[s].flags, FL_SLOW);
#ifdef RSX_EXPERIMENT
    // why (bit k = reason k: 1 more than 128 symbols in a slot, 2 side buffer full, 3 round
    // limit, 4 re-decode overflow, 5 exit changed by a repair, 6 look-back 0 gave up, 7
    // invalid code, 8 staging capacity / rows, 9 look-back 1 gave up)
    atomicOr(&a.results[s].stat_why, 1u << F.misc[M_SLOW]);
#endif
  }
  lds_barrier();
  LF_STAMP(13);
  LF_STAMP(14);
  // 9. copy-out
  if (fits && any_out && !(LF_ABLATE & 3u))
    lf_copy_out2<N, NK>(F, a, S, base, lim, sb, r0, j, walk0,
                        DIFF ? reinterpret_cast<uint8_t*>(a.diffs + S.diff_offset) : a.out_base + S.img_offset,
                        nko);
#ifdef RSX_EXPERIMENT
  // K0's count of the slot against this kernel's (the first slot of the stream that differs)
  if (j >= 1 && a.sub_sums && own_bits != 0u) {
    const uint2 k = a.sub_sums[gsub];
    if ((k.x & 0xFFFFu) != rec_cn(my_rec_final) &&
        atomicCAS(&a.results[s].pad3[0], 0u, 0x08000000u | (lb << 8) | uint32_t(j)) == 0u) {
      a.results[s].pad3[1] = k.x;
      a.results[s].pad3[2] = k.y | (rec_cn(my_rec_final) << 24);
    }
  }
#endif
  // records the per-stream bookkeeping kernels read (lj_scan_kernel, lj_consumed_kernel)
  if (j >= 1)
    a.sub_state[gsub] = rec_st(my_rec_final) | (rec_cn(my_rec_final) << 16);
  if (j == 0) {
    a.block_start[b] = entry_final;
    a.block_exit[b] = published_exit;
    a.block_base0[b] = base;
    a.block_sum[b] = cnt_wg;
    a.block_flags[b] = 0;
    a.block_psum[b] = S_wg;
  }
  LF_STAMP(15);
}

template <int N, int TM>
void launch_fast_one(const LjArgs& a, const FastLaunch& f, hipStream_t s, KernelTimer* timer) {
  if (!f.present[TM][N])
    return;
  const bool probe = (a.fast_level_mask & (a.fast_level_mask - 1u)) != 0u; // (more than one level)
  for (uint32_t lv = 0; lv < 3; ++lv) {
    if (!((a.fast_level_mask >> lv) & 1u))
      continue;
    if (a.dev_layout)
      hipLaunchKernelGGL((lj_fast_kernel<N, TM, 2>), dim3(a.blk_n), dim3(LJ_T),
                         a.fast_lds_lv[lv], s, a, a.fast_lds_lv[lv], lv);
    else if (probe)
      hipLaunchKernelGGL((lj_fast_kernel<N, TM, 1>), dim3(a.blk_n), dim3(LJ_T),
                         a.fast_lds_lv[lv], s, a, a.fast_lds_lv[lv], lv);
    else
      hipLaunchKernelGGL((lj_fast_kernel<N, TM, 0>), dim3(a.blk_n), dim3(LJ_T),
                         a.fast_lds_lv[lv], s, a, a.fast_lds_lv[lv], lv);
    if (timer)
      timer->mark(lv == 0 ? "lj_fast_kernel" : (lv == 1 ? "lj_fast_kernel(3/CU)" : "lj_fast_kernel(2/CU)"));
  }
}

} // namespace

// LDS bytes of a launch whose workgroups deliver up to `samples` symbols each (0 if no
// allocation that keeps two workgroups on a CU holds them)
uint32_t ljpeg_fast_lds_for(uint64_t samples) {
  uint64_t need = 2 * samples + LF_TAIL_BYTES + LF_STAGE_BASE + 16;
  if (need < LF_LDS_MIN)
    need = LF_LDS_MIN;
  need = (need + 1279) / 1280 * 1280;
  if (need < 40960)
    need = 40960; // (four workgroups per CU take 40 KB each anyway)
#ifdef RSX_LF_MIN_LDS // (experiment: fewer resident workgroups, for the residency slope)
  if (need < RSX_LF_MIN_LDS)
    need = RSX_LF_MIN_LDS;
#endif
  return need <= 64 * 1024 ? uint32_t(need) : 0u; // (more needs hipFuncSetAttribute)
}

uint32_t ljpeg_fast_stage_cap(uint32_t lds_bytes) {
  return (lds_bytes - LF_TAIL_BYTES - LF_STAGE_BASE - 16u) / 2u;
}

// (streams that leave differences: <1, 0, MODE, true>)
static void launch_fast_diffs(const LjArgs& a, const FastLaunch& f, hipStream_t s, KernelTimer* timer) {
  if (!f.diffs)
    return;
  const bool probe = (a.fast_level_mask & (a.fast_level_mask - 1u)) != 0u;
  for (uint32_t lv = 0; lv < 3; ++lv) {
    if (!((a.fast_level_mask >> lv) & 1u))
      continue;
    if (a.dev_layout)
      hipLaunchKernelGGL((lj_fast_kernel<1, 0, 2, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv], s, a,
                         a.fast_lds_lv[lv], lv);
    else if (probe)
      hipLaunchKernelGGL((lj_fast_kernel<1, 0, 1, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv], s, a,
                         a.fast_lds_lv[lv], lv);
    else
      hipLaunchKernelGGL((lj_fast_kernel<1, 0, 0, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv], s, a,
                         a.fast_lds_lv[lv], lv);
    if (timer)
      timer->mark(lv == 0 ? "lj_fast_kernel<differences>" : "lj_fast_kernel<differences>(fewer/CU)");
  }
}

// (Nikon-type streams' pixels: <2, 0, MODE, false, true>)
static void launch_fast_nk(const LjArgs& a, const FastLaunch& f, hipStream_t s, KernelTimer* timer) {
  if (!f.nk)
    return;
  const bool probe = (a.fast_level_mask & (a.fast_level_mask - 1u)) != 0u;
  for (uint32_t lv = 0; lv < 3; ++lv) {
    if (!((a.fast_level_mask >> lv) & 1u))
      continue;
    if (a.dev_layout)
      hipLaunchKernelGGL((lj_fast_kernel<2, 0, 2, false, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv],
                         s, a, a.fast_lds_lv[lv], lv);
    else if (probe)
      hipLaunchKernelGGL((lj_fast_kernel<2, 0, 1, false, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv],
                         s, a, a.fast_lds_lv[lv], lv);
    else
      hipLaunchKernelGGL((lj_fast_kernel<2, 0, 0, false, true>), dim3(a.blk_n), dim3(LJ_T), a.fast_lds_lv[lv],
                         s, a, a.fast_lds_lv[lv], lv);
    if (timer)
      timer->mark(lv == 0 ? "lj_fast_kernel<nikon-type>" : "lj_fast_kernel<nikon-type>(fewer/CU)");
  }
}

void ljpeg_launch_fast(const LjArgs& a, const FastLaunch& f, hipStream_t s, KernelTimer* timer) {
  launch_fast_diffs(a, f, s, timer);
  launch_fast_nk(a, f, s, timer);
  launch_fast_one<1, 0>(a, f, s, timer);
  launch_fast_one<2, 0>(a, f, s, timer);
  launch_fast_one<3, 0>(a, f, s, timer);
  launch_fast_one<4, 0>(a, f, s, timer);
  launch_fast_one<2, 1>(a, f, s, timer);
  launch_fast_one<4, 1>(a, f, s, timer);
  launch_fast_one<2, 2>(a, f, s, timer);
  launch_fast_one<3, 2>(a, f, s, timer);
  launch_fast_one<4, 2>(a, f, s, timer);
}

// The LUT of the fast loops from the 11-bit table of the general ones.
void ljpeg_build_fast_table(const TabLds& t, uint2* out, uint32_t* zinfo) {
  *zinfo = 0;
  for (uint32_t i = 0; i < 1024; ++i) {
    const uint32_t ea = reinterpret_cast<const uint16_t*>(t.lut)[2 * i * (sizeof(LutEntry) / 2)];
    const uint32_t eb =
        reinterpret_cast<const uint16_t*>(t.lut)[(2 * i + 1) * (sizeof(LutEntry) / 2)];
    const uint32_t cl = ea & 31u, ssss = (ea >> 5) & 31u, total = ea >> 10;
    const bool plain = ea != 0 && cl <= 10u && ssss < 16u && total == cl + ssss && total >= 1u &&
                       total <= 26u;
    if (plain) {
      out[i] = make_uint2((32u - total) | (total << 5), (1u << ssss) - 1u);
      if (ssss == 0)
        *zinfo = cl | ((i >> (10u - cl)) << 8);
    } else {
      // the warm-up just moves on: by the symbol's length if an 11-bit code says so
      uint32_t adv = 16;
      if (ea != 0 && (ea >> 10) >= 1u)
        adv = ea >> 10;
      else if (eb != 0 && (eb >> 10) >= 1u)
        adv = eb >> 10;
      if (adv > 63u)
        adv = 63u;
      out[i] = make_uint2(0x80000000u | (adv << 5), 0u);
    }
  }
}

// The 2-byte form of the same table for streams with a table per phase (lf_step_pt).
void ljpeg_build_fast_table16(const uint2* t8, uint16_t* out) {
  for (uint32_t i = 0; i < 1024; ++i) {
    const uint2 e = t8[i];
    if (e.x & 0x80000000u) {
      out[i] = uint16_t(LF_PT_SPECIAL);
    } else {
      const uint32_t ssss = uint32_t(__builtin_popcount(e.y)); // (e.y = 2^SSSS - 1, SSSS <= 15)
      out[i] = uint16_t((e.x & 0x7FFu) | (ssss << 11));
    }
  }
}

} // namespace rsx
