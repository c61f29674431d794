// rsx_ljpeg_dev.h -- constants and device-visible structures shared by the
// translation units of the lossless-JPEG family pipeline:
//   rsx_ljpeg.hip         un-stuffing, synchronisation, scans, the host plan; the legacy
//                         decode into int16 differences (K4, tail) for the stream kinds
//                         the fused path does not cover
//   rsx_ljpeg_direct.hip  fused decode + predictor reconstruction (row edges, row
//                         offsets, decode straight into the image)
//   rsx_ljpeg_recon.hip   legacy reconstruction from differences (K5 seeds, K6 row
//                         scans) and the Nikon / Pentax / Sony kernels
#pragma once

#include "rsx_internal.h"

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace rsx {

// ---------------------------------------------------------------------------
// Geometry constants
// ---------------------------------------------------------------------------
constexpr int LJ_T = 256;             // lanes per workgroup = slots per workgroup
#ifndef RSX_LJ_P
#define RSX_LJ_P 64
#endif
constexpr int LJ_P = RSX_LJ_P;        // physical bytes per subsequence
constexpr int LJ_PW = LJ_P / 4;       // dwords per subsequence
constexpr int LJ_OWN = LJ_T - 1;      // owned slots (slot 0 = warm-up)
constexpr int LJ_R = LJ_OWN * LJ_P;   // bytes of stream owned by one workgroup
constexpr int LJ_BW = LJ_PW + 4;      // compacted slot capacity (dwords)
constexpr int LJ_IMG_U4 = (LJ_BW + 1) * LJ_T / 4; // per-workgroup un-stuffed image: B + ob[], in uint4
#ifndef RSX_LJ_WARM
#define RSX_LJ_WARM 512
#endif
constexpr uint32_t LJ_WARM = RSX_LJ_WARM;     // warm-up bits decoded ahead of a slot for its start guess

// Ablation switches of the experiment builds (rawspeed_amd/build.py build_variant with
// -DRSX_EXPERIMENT -DRSX_ABLATE=<bits>; compile-time constants, 0 in the shipped
// library).  K1: 4 = no recorded pass, 16 = no warm-up, 32 = no re-decode rounds.
// K5a: 64 = no walks.  K4d: 1 = nothing after the decode of a burst, 2 = no decode
// loop, 1024 = no 16-byte stores, 4096 = no general store path, 8192 = no tail
// stores, 16384 = no initial cursor.
#if defined(RSX_EXPERIMENT) && defined(RSX_ABLATE)
constexpr uint32_t LJ_ABLATE = RSX_ABLATE;
#else
constexpr uint32_t LJ_ABLATE = 0;
#endif

constexpr uint32_t ST_OFF_MASK = 63u;
constexpr uint32_t ST_PHASE_SHIFT = 6;
constexpr uint32_t ST_ERR = 1u << 9;
constexpr uint32_t ST_MASK = 0xFFFFu;

constexpr uint32_t NO_CODE = 0xFFFFFFFFu;

// flags in LjResult::flags
constexpr uint32_t FL_UNCONVERGED = 1u;
// a stream of the fused path whose symbols run past the end of its data (damaged or
// truncated input): it is redone by the legacy kernels, which know the reference's
// end-of-stream semantics (lj_tail_kernel)
constexpr uint32_t FL_NEED_LEGACY = 2u;
// some workgroup of the stream met periodic data (constant image regions)
constexpr uint32_t FL_PERIODIC = 4u;
// a stream of the single-pass path (rsx_ljpeg_fast.hip) that it could not finish --
// periodic data beyond its round limit, a subsequence of more than LF_MAXSYM symbols,
// an invalid code, an entry state that moved after it was published: the stream is
// redone by the multi-kernel pipeline (LjArgs::pass == 1)
constexpr uint32_t FL_SLOW = 8u;

// diff_offset of a fused-path stream before its difference scratch exists (it is set up
// on first use, LJpegPlan::legacy_fallback_ready): the legacy kernels must not touch such a
// stream even when it is flagged FL_NEED_LEGACY -- in a plan that mixes legacy-route and
// fused-path streams they run in the FIRST pass too, and would write the damaged stream's
// differences over a healthy stream's region.
constexpr uint64_t LJ_NO_DIFFS = ~uint64_t(0);

// The code table as the kernels keep it in LDS.  RSX_LUT_DIFF: a LUT entry also
// carries, in its high half, the DIFFERENCE the symbol stands for whenever the whole
// symbol -- code and difference bits -- lies inside the LUT_BITS index bits (total <=
// LUT_BITS; sensor data: nearly always), so that the loops which need differences (the
// recorded synchronisation pass, the final decode) read them instead of computing the
// JPEG "EXTEND"; longer symbols take the computed path, wave-uniformly.
// OFF: the 8 KB table costs the synchronisation kernel its sixth workgroup per CU, which
// is worth more than the instructions saved (production builds side by side, one GPU
// session: cfg 3 1.274 -> 1.232 ms, uniform-random 14-bit 1.09 -> 0.96, clipped
// highlights 2.48 -> 2.25 with the table off).  Kept for experiments.
#ifndef RSX_LUT_DIFF
#define RSX_LUT_DIFF 0
#endif
#if RSX_LUT_DIFF
typedef uint32_t LutEntry;
#else
typedef uint16_t LutEntry;
#endif

struct alignas(16) TabLds {
  LutEntry lut[LUT_SIZE];
  uint32_t max_code[18];
  uint16_t val_offset[18];
  uint8_t values[RSX_MAX_CODE_VALUES];
  uint8_t max_len;
  uint8_t fix16;
  uint8_t zero_sym_bits;
  uint8_t las;
};
static_assert(sizeof(TabLds) % 16 == 0, "TabLds must be 16-byte sized");

struct Cr2Strip {
  uint32_t x0, w, y0, h;
  uint64_t first_sample;
};

struct LjStreamDev {
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t diff_offset; // int16 index into the difference scratch (multiple of 8)
  uint64_t needed;      // symbols the reference decodes for this stream
  uint64_t img_offset;
  uint32_t img_pitch;
  uint32_t first_block;
  uint32_t n_blocks;
  uint32_t first_subseq;
  uint32_t table_base;
  uint32_t n_tables;
  uint32_t period;
  uint32_t n_comp;
  uint8_t tab_of_phase[8];
  uint16_t init_pred[4];
  uint8_t seed_pos[4]; // first sample of component c inside a stream row
  uint8_t raw;         // 1: plain MSB bit stream (BitStreamerMSB): no FF00 un-stuffing,
                       //    no markers, position budget of 8 bytes instead of 16
  uint8_t start_bit;   // first symbol starts this many bits into the stream (0..7)
  uint8_t las;         // table values are Nikon "lossy after split" (len | shl << 4)
  uint8_t pair;        // HasselbladDecompressor: a symbol is [code1][code2][bits1][bits2]
                       // (2 differences); the stream is MSB32 (LE 32-bit words, MSB first)
  uint8_t no_vertical; // every stream row starts from init_pred (no seed chain)
  uint8_t direct;      // != 0: fused decode + reconstruction (rsx_ljpeg_direct.hip); the
                       //       value is the number of interleaved components (1, 2 or 4)
  uint8_t sync_lut11;  // its table has no search path past the LUT (an explicit 11-bit table):
                       // the synchronisation kernels must not use their 10-bit LUT for it
  uint8_t fast;        // != 0: the single-pass kernel decodes it (rsx_ljpeg_fast.hip): 1 one
                       // table, 2 two tables alternating symbol by symbol, 3 a table per phase
  uint8_t tab_period;  // fast == 3: the period of tab_of_phase over the components (2, 3 or 4: A B A B
                       // is 2) -- the phase of a parse state is the symbol index mod this
  uint8_t fast_diffs;  // fast != 0: the single-pass kernel leaves the stream's DIFFERENCES (int16, stream
                       // order, at diff_offset) for the legacy reconstruction kernels instead of pixels
                       // -- Nikon-type predictors, Pentax, Canon sRaw groups: one table (round 6)
  uint8_t fast_nk;     // fast != 0: a Nikon-type stream (kind 2: stride-2 left predictor, the first pair of
                       // a row predicted from the row TWO above, NikonDecompressor.cpp:518-560 /
                       // PentaxDecompressor.cpp:155-177) whose PIXELS the single-pass kernel writes, curve
                       // and dither in its copy-out; init_pred holds pUp by STREAM-row parity
                       // ([2 * (r & 1) + c]).  The kernel's sums are mod 2^16, the reference's plain ints:
                       // a value with bit 15 set (Pentax: one that does not fit range_bits) gives the
                       // stream to the legacy route, whose ints say what it really was
  uint8_t pad_fast_[1];
  uint32_t rows;
  uint32_t row_samples;
  uint32_t first_row; // global stream-row index
  uint32_t kind;      // 0 LJPEG, 1 CR2
  uint32_t mcu_w, mcu_h, out_x, out_y, keep_samples;
  uint32_t scan_samples; // samples of a row that take part in reconstruction
  uint32_t n_strips;
  uint32_t strip_base;
  uint32_t job;
  uint64_t raw_limit;  // raw streams: 1 + last bit offset a symbol may start at (0 = derive)
};

// Per-stream parameters of NikonDecompressor streams (kind 2), indexed like streams[].
struct NkStreamDev {
  int32_t p_up[4];       // pUp[row & 1][col & 1] at [2 * (row & 1) + (col & 1)]
  const int32_t* pup_in; // non-null: read the initial pUp from here instead (rows after the split)
  uint32_t uncorrected;  // 1: store clampBits(pred, 15) as is
  uint32_t table_off;    // first entry of this stream's dither table in nk_tables
  uint32_t rowpow_off;   // first entry of this stream's row powers in nk_rowpow
  uint32_t pentax;       // != 0: PentaxDecompressor (.cpp:152-176) / SamsungV1 (.cpp:125-137):
                         //    no clamp; = number of bits a value may have, more is
                         //    RSX_ERR_VALUE_RANGE
  uint64_t seed_offset;  // byte offset (from in_base) of the job's first input byte
  uint32_t sony;         // != 0: SonyArw1Decompressor (.cpp:59-93): stream row r = image column
                         //    W-1-r (even rows, then odd rows), one predictor through all rows,
                         //    values outside 0..4095 are RSX_ERR_VALUE_RANGE
  uint32_t colpow_off;   // fast_nk with dither: first entry in nk_rowpow of 15700^x mod m, x < row_samples
};

struct LjResult {
  uint32_t marker_pos; // first FFxx (xx != 0) in the stream, 0xFFFFFFFF = none
  uint32_t status;
  uint32_t flags;
  uint32_t avail_lo;   // symbols that start before the end of data
  uint32_t last_slot;  // stream-relative subsequence of the last needed symbol
  uint32_t last_pos;   // its bit offset inside the compacted subsequence
  uint32_t consumed;
  uint32_t tail_used;  // 1: the tail kernel delivered the last symbols
  uint32_t stat_why;   // statistics (experiment builds): why the single-pass kernel gave up
  uint32_t pad3[3];
  uint32_t last_c_lo;  // un-stuffed bit offset of the last symbol (tail path)
  uint32_t last_c_hi;
  uint32_t stat_rounds; // statistics: re-decode rounds summed over workgroups
  uint32_t stat_redo;   // statistics: slots re-decoded
  uint32_t stat_stitch; // statistics: workgroups re-converged by the stitch kernel
  uint32_t end_lo;      // raw streams: bit offset just past the last needed symbol
  uint32_t end_hi;
  uint32_t pad2;        // statistics (RSX_DEBUG): most re-decode rounds of any workgroup << 16 | its index
};

struct LjArgs {
  const uint8_t* in_base;
  uint8_t* out_base;
  const LjStreamDev* streams;
  const TabLds* tables;
  const uint32_t* block_stream;
  const Cr2Strip* strips;
  uint32_t* sub_state;       // per subsequence: exit state | symbols << 16
  uint16_t* sub_start;       // per subsequence: the state it was decoded from
  uint2* sub_sums;           // per subsequence: sums of its differences by relative phase
  uint32_t* sub_first;       // per subsequence: symbols before it inside its workgroup
  uint2* sub_psum;           // per subsequence: running sums P before it inside its
                             // workgroup, by phases relative to the workgroup's first symbol
  uint32_t* block_start;
  uint32_t* block_exit;
  uint32_t* block_flags;     // per workgroup: 1 = gave up on its re-decode rounds,
                             //                2 = block_tf holds its transfer function
  uint16_t* block_tf;        // per workgroup: exit state for each of the 32 entry offsets
                             //                (0xFFFF = unknown), periodic data only
  uint32_t* block_sum;
  uint32_t* block_base;
  uint2* block_psum;         // per workgroup: sums of its differences, phases relative to
                             // its first symbol
  uint2* block_pbase;        // per workgroup: running sums P (absolute components) before
                             // its first symbol
  uint4* row_edge;           // per stream row: P before its first symbol (x, y) and at its
                             // first MCU (z, w)
  uint32_t* block_drops;     // stuffing bytes dropped inside each workgroup's region
  uint32_t* block_drop_base; // exclusive prefix of block_drops within the stream
  uint4* unstuffed;          // per workgroup: its LDS image of un-stuffed slots (LJ_BW*LJ_T dwords)
  LjResult* results;
  LjResult* results_next;    // the next run's set (cleared by lj_unstuff_kernel); nullptr: none
  int16_t* diffs;
  uint16_t* vseed;           // per stream row, 4 x u16: predictor seeds (legacy path) /
                             // row offsets O(r, c) (fused path)
  uint32_t n_streams;
  uint32_t total_rows;
  // NikonDecompressor streams
  const NkStreamDev* nk;
  const uint32_t* nk_tables; // dither tables: base | delta << 16 per 15-bit value
  const uint32_t* nk_rowpow; // 15700^(y * W) mod (15700 * 2^16 - 1) per output row
  int32_t* nk_pup;           // [stream][4]: pUp after the stream's last row
  uint16_t* transfer;        // [workgroup][512]: exit state per entry state (fallback path)
  // single-pass path (rsx_ljpeg_fast.hip)
  const uint2* fast_tabs;    // [table][1024]: the 10-bit LUT of the single-pass loops
  const uint16_t* fast_tabs16; // [table][1024]: its 2-byte form (streams with a table per phase)
  unsigned long long* lb;    // [workgroup][LF_LB_WORDS]: look-back records (zeroed by K0)
  uint32_t* tickets;         // [3][4]: workgroup tickets of the single-pass launches by
                             // LDS level and components (zeroed by K0)
  uint32_t* fast_level;      // [2]: the LDS level K0 chose for this run (see fast_lds_lv), by
                             // run parity; [2 + parity]: the level it would have needed.  (Not next to the tickets: 15 000 workgroups'
                             // atomics and loads on one cache line take 7 ns each, in turn.)
  uint32_t fast_lds;         // LDS bytes of the single-pass launches (staging capacity)
  // The single-pass kernel stages all samples of a workgroup in LDS.  How many that are
  // depends on the DATA (a constant region has six times the symbols per byte of sensor
  // noise), so the kernel is launched once per LDS level -- 4, 3 and 2 workgroups per CU --
  // and K0, which counts the symbols of every workgroup on the way, says which of the
  // launches does the work; the workgroups of the others leave at once.
  uint32_t fast_lds_lv[3];   // LDS bytes per level (ascending; equal entries = level absent)
  uint32_t fast_cap_lv[3];   // samples a workgroup can stage at that level
  uint32_t run_parity;       // which of the two level words this run uses (K0 clears the other)
  uint32_t fast_level_mask;  // bit l: level l is launched in this run.  (An empty launch costs
                             // 7 us; once the host has seen which level a plan's data needs --
                             // fast_level[2 + parity], fetched with the results -- it launches
                             // that one only.  K0 picks among the launched levels; workgroups
                             // that do not fit theirs send the stream to the multi-kernel pipeline.)
  const uint32_t* fast_z;    // [table]: code of the zero difference: length | code << 8 (0: none)
  uint32_t guess_slots;      // slots K0 parses for a start guess (2 or 3)
  const uint4* fast_order;   // [ticket]: the workgroup's (block, stream, table) -- the streams'
                             // blocks interleaved, each stream's in order
  unsigned long long* k0w;   // [workgroup]: what K0 knows about its symbols (lj_unstuff_kernel):
                             // count | own estimate of its entry state, "uncertain" | true entry
  uint32_t* block_base0;     // [workgroup]: the symbol base the single-pass kernel worked from
  uint32_t* k0p;             // [2][workgroup + 1]: K0's look-back over the workgroups' symbol counts mod N
                             // (streams with a table per phase: the phase a workgroup starts in), one set
                             // per run parity, each run clears the other one's
  uint32_t pt_np;            // table-per-phase plans: the most phases a stream has (K0's LDS layout)
  uint32_t* k0e;             // [workgroup]: K0's hand-over of entry states between its workgroups:
                             // 0x8000 | run parity << 14 | the state the predecessor's chain ends in
  uint32_t k0_chain;         // != 0: K0 runs in block order and hands entry states over
  unsigned long long* dbg;   // experiment builds: [workgroup][16] phase time stamps
  uint32_t fast_uniform_nb;  // != 0: every stream of the plan has this many workgroups, so the
                             // single-pass kernel works its (stream, block) out of its block
                             // index instead of loading fast_order's entry (one dependent
                             // round trip less in front of its image loads)
  uint32_t fast_rotate;      // != 0: the streams' turn inside a group of n_streams tickets
                             // rotates from group to group (see ljpeg_plan_create)
  uint32_t dev_layout;       // != 0: streams[], block_stream[] and fast_order[] are REWRITTEN by
                             // a kernel in every run (restart intervals laid out on the device,
                             // lj_dri_layout_kernel): the kernels that read them through the
                             // scalar cache invalidate it first (lj_fresh_scalars)
  uint32_t fuse_consumed;    // != 0: lj_scan_kernel does lj_consumed_kernel's work as well
  uint32_t blk0, blk_n;      // the blocks of THIS launch of K0 / the single-pass kernel: [blk0, blk0 + blk_n)
                             // -- all of them as a rule; a part of them when a host-pointer call runs a
                             // stream in chunks, its upload and download under the decode (round 6)
  uint32_t n_blocks_plan;    // the plan's blocks (the stride of per-plan arrays that gridDim.x was)
  uint32_t pass;             // 0: first pass; 1: the multi-kernel pipeline redoes FL_SLOW
                             // streams; 2: its streams of both passes
};

// words of a workgroup's look-back record (8-byte granules, each self-validating):
// [0] entry / exit state, symbols, inclusive symbol base; [1..4] LOCAL transfer of the
// predictor state (a, v; two words each for 4 components); [5..8] the inclusive state
#ifdef RSX_LF_LB16 // (experiment: LOCAL / inclusive pairs of a record as 16-byte loads)
constexpr int LF_LB_WORDS = 10;
#else
constexpr int LF_LB_WORDS = 9;
#endif

// Nothing invalidates a CU's scalar data cache between two kernels that follow one another on
// a stream without the host in between (measured in round 3 on the LDS-level word): a kernel
// that reads, through scalar loads, words another kernel has rewritten since the cache last saw
// them could see the old ones.  Plans laid out on the host never rewrite such words between
// kernels; plans laid out on the device (restart intervals, lj_dri_layout_kernel) do, every
// run: their K0, single-pass kernel and scan are INSTANTIATIONS OF THEIR OWN (INV) whose every
// wavefront drops the scalar cache before its first load -- deterministic, whatever else
// occupies the chip, and no instruction in the kernels of the other plans.  (Until round 6: a
// kernel of 4096 one-wavefront workgroups behind the layout, "a wavefront on every CU" -- true
// on an idle chip only.  As a run-time `if` at the top of the shared instantiations the
// invalidation cost EVERY plan 1-2 %: the asm statement kept the compiler from batching the
// kernels' first loads.)
template <bool INV>
__device__ __forceinline__ void lj_fresh_scalars() {
  if constexpr (INV)
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// Which streams a kernel of the multi-kernel pipeline works on.  First pass: every
// stream the single-pass kernel does not take.  Second pass (launched when the first one
// left FL_SLOW streams): exactly those.
__device__ __forceinline__ bool lj_pipeline_takes(const LjArgs& a, uint32_t s,
                                                  const LjStreamDev& S) {
  if (S.fast == 0)
    return a.pass != 1;
  // (pass 2: the host-driven convergence rounds, after both passes: everything the
  // multi-kernel pipeline owns by now)
  return a.pass != 0 && (a.results[s].flags & FL_SLOW) != 0;
}
// ... the per-stream bookkeeping kernels (scan, consumed) that also serve the
// single-pass streams in the first pass
__device__ __forceinline__ bool lj_bookkeeping_takes(const LjArgs& a, uint32_t s,
                                                     const LjStreamDev& S) {
  return a.pass == 0 || lj_pipeline_takes(a, s, S);
}

// whether the legacy route (int16 differences + K5 / K6, lj_tail_kernel's end-of-stream
// rules) takes stream s in this launch
__device__ __forceinline__ bool lj_legacy_takes(const LjArgs& a, uint32_t s,
                                                const LjStreamDev& S) {
  if (!lj_pipeline_takes(a, s, S))
    return false;
  if (!S.direct)
    return true;
  return (a.results[s].flags & FL_NEED_LEGACY) != 0 && S.diff_offset != LJ_NO_DIFFS;
}

// ... whether the reconstruction kernels (seeds, row scans) take it: the legacy route's streams, and in
// the first pass the streams whose differences the single-pass kernel has just left (fast_diffs)
// -- unless it gave the stream up: then the second pass decodes and reconstructs it
__device__ __forceinline__ bool lj_recon_takes(const LjArgs& a, uint32_t s, const LjStreamDev& S) {
  if (lj_legacy_takes(a, s, S))
    return true;
  return S.fast != 0 && S.fast_diffs != 0 && a.pass == 0 && !(a.results[s].flags & FL_SLOW);
}
// (the Nikon-type kernels take every stream of their kind: not the ones given up in the first pass,
// and of the streams whose pixels the single-pass kernel writes (fast_nk) only those it gave up on,
// in the pass that redoes them)
__device__ __forceinline__ bool lj_recon_skips_given_up(const LjArgs& a, uint32_t s, const LjStreamDev& S) {
  if (S.fast != 0 && S.fast_nk != 0)
    return !(a.pass != 0 && (a.results[s].flags & FL_SLOW) != 0);
  return S.fast != 0 && S.fast_diffs != 0 && a.pass == 0 && (a.results[s].flags & FL_SLOW) != 0;
}

constexpr int VS_T = 1024; // lanes of the per-stream seed kernels

// what the reconstruction launch needs to know about the plan
struct ReconLaunch {
  uint32_t n_streams = 0;
  uint32_t total_rows = 0;
  bool comp_present[7] = {}; // [1..4] interleaved n_comp; [5], [6]: sRaw groups of 4, 6
  bool any_nikon = false;
  bool any_sony = false;
};

// K5 + K6 (and their Nikon / Pentax counterparts) on `stream`
void ljpeg_launch_reconstruct(const LjArgs& a, const ReconLaunch& r, hipStream_t stream);

// optional per-kernel timing of a plan run: an event after every launch
struct KernelTimer {
  static constexpr int MAX = 48;
  hipEvent_t ev[MAX + 1] = {};
  const char* name[MAX] = {};
  int n = 0;       // launches recorded in this run
  int created = 0; // events created so far
  hipStream_t stream = nullptr;
  void begin(hipStream_t s);
  void mark(const char* kernel); // the launch just issued
};

// what the fused decode launch needs to know about the plan
struct DirectLaunch {
  uint32_t n_streams = 0;
  uint32_t total_blocks = 0;
  uint32_t total_rows = 0;
  int max_tables = 1;
  bool present[2][5] = {}; // [several tables][components]
};
void ljpeg_launch_direct(const LjArgs& a, const DirectLaunch& d, hipStream_t stream,
                         KernelTimer* timer);

// the single-pass path (rsx_ljpeg_fast.hip)
struct FastLaunch {
  uint32_t total_blocks = 0;
  bool present[3][5] = {}; // [one table / two alternating / a table per phase][components]
  bool diffs = false;      // some stream leaves differences (fast_diffs): the <1, 0, ., true> instantiation
  bool nk = false;         // some Nikon-type stream's pixels (fast_nk): the <2, 0, ., false, true> instantiation
};
void ljpeg_launch_fast(const LjArgs& a, const FastLaunch& f, hipStream_t stream,
                       KernelTimer* timer);
// the 10-bit LUT of the single-pass loops for one table (1024 entries)
void ljpeg_build_fast_table(const TabLds& t, uint2* out, uint32_t* zinfo);
// ... and its 2-byte form (shift | total << 5 | SSSS << 11; bit 15: special)
void ljpeg_build_fast_table16(const uint2* t8, uint16_t* out);
// A workgroup's table word (fast_order[].z, or put together from the stream's record): the
// stream's first table (16 bits) | its table of phase 0, 1, 2, 3 (4 bits each at 16, 20, 24, 28).
// One table: phase 0's is 0.  Two alternating tables: phase 0 = the even symbols', 1 = the odd ones'.
__host__ __device__ __forceinline__ uint32_t lf_table_word(uint32_t base, uint32_t t0, uint32_t t1,
                                                           uint32_t t2, uint32_t t3) {
  return (base & 0xFFFFu) | ((t0 & 15u) << 16) | ((t1 & 15u) << 20) | ((t2 & 15u) << 24) |
         ((t3 & 15u) << 28);
}
uint32_t ljpeg_fast_lds_for(uint64_t samples_per_workgroup);
uint32_t ljpeg_fast_stage_cap(uint32_t lds_bytes);
constexpr int LF_TICKET_WORDS = 32; // [2][3][4] tickets: [two tables][LDS level][components]

} // namespace rsx
