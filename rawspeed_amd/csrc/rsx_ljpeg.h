// rsx_ljpeg.h -- host interface of the lossless-JPEG decode pipeline
// (implementation + kernels: rsx_ljpeg.hip).
#pragma once

#include "rsx_internal.h"

#include <vector>

namespace rsx {

// NikonDecompressor jobs (StreamGeom::kind == 2): what the reconstruction
// kernels need besides the entropy decode.
struct NikonIn {
  int32_t p_up[4] = {0, 0, 0, 0}; // pUp[row & 1][col & 1] at [2 * (row & 1) + (col & 1)]
  const int32_t* pup_in = nullptr; // device pointer overriding p_up (rows after a split)
  bool uncorrected = true;
  bool pentax = false;    // PentaxDecompressor / SamsungV1: predictors start at 0, a
                          // value that does not fit range_bits bits is an error
  int range_bits = 16;
  bool sony = false;      // SonyArw1Decompressor: a stream row is an image column, ONE
                          // predictor runs through all rows, values must be 0..4095
  int split = 0;          // rows >= split use table_after_split (0 = none)
  int height = 0;         // image rows
  std::vector<uint32_t> dither; // 32768 x (base | delta << 16); empty if uncorrected
  rsx_huff_table table_after_split{};
  uint64_t seed_offset = 0; // first byte of the job's input (the 24-bit dither seed)
};

struct LJpegJobIn {
  int status = RSX_OK;    // validation result; failed jobs are skipped
  StreamGeom geom;        // flattened geometry (rsx_host.cpp)
  const rsx_huff_table* tables = nullptr;
  int n_tables = 0;
  int rows_per_restart_interval = 0; // LJPEG only; 0 = no restart markers
  int frame_h = 0;
  NikonIn nikon;          // kind 2 only
  // != nullptr: the one table of the job is not a canonical JPEG code; it is
  // given as the reference's own (encLen, diffLen) pairs (SamsungV1)
  const uint8_t* explicit_enc_len = nullptr;
  const uint8_t* explicit_diff_len = nullptr;
  int explicit_n = 0;
  int explicit_bits = 10; // index width of that table (<= LUT_BITS)
};

struct LJpegPlan;

int ljpeg_plan_create(rsx_ctx* ctx, const std::vector<LJpegJobIn>& jobs,
                      LJpegPlan** out);
struct KernelTimer; // rsx_ljpeg_dev.h: an event after every launch of the run
int ljpeg_plan_run(LJpegPlan* plan, const void* in_dev, void* out_dev,
                   hipStream_t stream, KernelTimer* timer);
// (continue_timer: a child plan's launches go on in the parent's kernel table)
int ljpeg_plan_run_(LJpegPlan* plan, const void* in_dev, void* out_dev, hipStream_t stream,
                    KernelTimer* timer, bool continue_timer);
int ljpeg_plan_results(LJpegPlan* plan, hipStream_t stream, bool ran,
                       int32_t* job_status, uint32_t* job_consumed);
void ljpeg_plan_destroy(LJpegPlan* plan);

// A run in chunks of the plan's blocks (round 6; rsx_api.hip, ljpeg_family_host): one stream of the
// single-pass kernel, each chunk's K0 + kernel queued behind the upload of its bytes, the pixels a
// chunk completes fetched while the next one decodes.
struct LjRegion {
  size_t byte0, bytes, row0, rows; // a rectangle of the image
};
bool ljpeg_plan_chunkable(const LJpegPlan* plan);
uint32_t ljpeg_plan_blocks(const LJpegPlan* plan);
int ljpeg_plan_run_begin(LJpegPlan* plan, const void* in_dev, void* out_dev, hipStream_t stream);
int ljpeg_plan_run_blocks(LJpegPlan* plan, hipStream_t stream, uint32_t blk0, uint32_t blk1);
int ljpeg_plan_run_end(LJpegPlan* plan, hipStream_t stream);
int ljpeg_plan_symbols_done(LJpegPlan* plan, hipStream_t stream, uint32_t blk_end, uint64_t* symbols);
void ljpeg_plan_region(const LJpegPlan* plan, uint64_t sym_lo, uint64_t sym_hi, std::vector<LjRegion>* out);
bool ljpeg_plan_single_pass_held(const LJpegPlan* plan);

struct LJpegPlanDeleter {
  void operator()(LJpegPlan* p) const { ljpeg_plan_destroy(p); }
};

} // namespace rsx
