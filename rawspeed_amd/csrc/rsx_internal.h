// rsx_internal.h -- shared declarations of the MI355X decompression core.
// Host-side structures only; device code lives in the .hip files.
#pragma once

#include "rsx.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace rsx {

// ------------------------------------------------------------------------
// Host-side validation = the reference constructors' checks, in their order
// (file:line cited at each check in rsx_host.cpp).
// ------------------------------------------------------------------------
int validate_unpack(const rsx_unpack_desc& d, const rsx_image& img,
                    size_t in_bytes);
int validate_unpack_f32(const rsx_unpack_desc& d, const rsx_image& img, size_t in_bytes);
int validate_unpack_variant(const rsx_unpack_variant_desc& d, const rsx_image& img,
                            size_t in_bytes);
int unpack_variant_bytes_per_line(const rsx_unpack_variant_desc& d, uint64_t* bpl);
int validate_ljpeg(const rsx_ljpeg_desc& d, const rsx_image& img);
int validate_cr2(const rsx_cr2_desc& d, const rsx_image& img);
int validate_nikon(const rsx_nikon_desc& d, const rsx_image& img);
int validate_pentax(const rsx_pentax_desc& d, const rsx_image& img);
int validate_samsung_v1(const rsx_samsung_v1_desc& d, const rsx_image& img);
int validate_sony_arw1(const rsx_image& img);
int validate_hasselblad(const rsx_hasselblad_desc& d, const rsx_image& img);
int validate_sraw(const rsx_sraw_desc& d, const rsx_image& in, const rsx_image& out);

// TableLookUp::setTable with dither (common/TableLookUp.cpp:50-84), 15-bit domain
void build_dither_table(const uint16_t* curve, int n, std::vector<uint32_t>* out);
int validate_huff_table(const rsx_huff_table& t);

// ------------------------------------------------------------------------
// Device-side Huffman table.  Canonical JPEG code (HuffmanCode.h:66-92) turned
// into (a) a direct LUT on the next LUT_BITS stream bits and (b) the Annex F
// maxcode/valptr arrays for longer codes.  Layout is ours, not the
// reference's 11-bit/int32 LUT (PrefixCodeLUTDecoder.h:86-92).
//   lut entry (u16): bits 0..4  code length (0 = code longer than LUT_BITS or
//                                invalid -> slow path)
//                    bits 5..9  SSSS category
//                    bits 10..15 total bits consumed by the symbol
//                                (code + difference bits, + the DNG-bug-16 skip)
// ------------------------------------------------------------------------
constexpr int LUT_BITS = 11;
constexpr int LUT_SIZE = 1 << LUT_BITS;

struct DeviceHuffTable {
  uint16_t lut[LUT_SIZE];
  // slow path (codes longer than LUT_BITS): for len in 1..16
  uint32_t max_code[18]; // 0xFFFFFFFF = no code of this length
  uint16_t val_offset[18]; // code - val_offset[len] = index into values
  uint8_t values[RSX_MAX_CODE_VALUES];
  uint8_t max_len;
  uint8_t fix16;
  // the symbol decoded from an all-zero bit stream (what the reference reads
  // past the end-of-stream marker, BitStreamerJPEG.h:155-179)
  uint8_t zero_sym_bits; // bits consumed by that symbol (0 = invalid code)
  // values are Nikon "lossy after split" codes (len | shl << 4; 16 = -32768):
  // NikonLASDecompressor::decodeDifference, NikonDecompressor.cpp:331-376
  uint8_t las;
};

void build_device_table(const rsx_huff_table& t, DeviceHuffTable* out, bool las = false);
// (encLen, diffLen) pairs in table-fill order -> the device LUT (no slow path:
// every code is at most `bits` <= LUT_BITS bits; the pairs tile a 2^bits table).
// diffLen 0xFF marks codes that are always an error.
void build_device_table_explicit(const uint8_t* enc_len, const uint8_t* diff_len, int n,
                                 DeviceHuffTable* out, int bits = 10);
int validate_las_table(const rsx_huff_table& t);

// ------------------------------------------------------------------------
// Job geometry handed to the LJPEG kernels.  One "stream" = one entropy-coded
// segment decoded with fresh predictors: a whole scan, or one restart
// interval.  Output mapping kinds:
//   LJPEG : stream sample (row r, s) -> MCU m = s / n_comp, comp c = s % n_comp
//           -> image (tile_y + mcu_h*r + c / mcu_w, tile_x_samples + mcu_w*m + c % mcu_w)
//           kept iff mcu_w*m + c%mcu_w < tile_w_samples (decodeRowN, LJpegDecompressor.cpp:200-250)
//   CR2   : stream group g -> vertical output strips (Cr2DecompressorImpl.h:431-465)
// ------------------------------------------------------------------------
constexpr int MAX_CR2_STRIPS = 64;

struct StreamGeom {
  // entropy-coded input of this stream, relative to the batch input base
  uint64_t in_offset;
  uint64_t in_bytes;
  // output image view, relative to the batch output base
  uint64_t img_offset;
  uint32_t img_pitch_bytes;
  // stream shape: `rows` stream rows of `row_samples` samples each
  uint32_t rows;        // rows actually decoded (tile rows / mcu_h, or frame rows)
  uint32_t row_samples; // frame_w * n_comp  (samples per stream row)
  uint32_t n_comp;      // predictor stride within a row
  uint32_t period;      // table cycle length (== n_comp for non-subsampled)
  uint8_t comp_of_phase[8]; // table slot for symbol index % period
  uint8_t pred_of_phase[8]; // predictor component for symbol index % period
  uint16_t init_pred[4];
  uint8_t seed_pos[4];      // first sample of predictor component c in a row
  uint32_t table_base;  // index of this job's first DeviceHuffTable
  // LJPEG mapping
  uint32_t kind;        // 0 LJPEG, 1 CR2, 2 Nikon (rows of row_samples at out_y, full width)
  uint32_t mcu_w, mcu_h;
  uint32_t out_x;       // first output sample column (cpp * tile_x)
  uint32_t out_y;       // first output row
  uint32_t keep_samples; // cpp * tile_w: samples of each output row that are stored
  // CR2 mapping: strips in stream order; strip k covers output sample columns
  // [x0, x0 + w) and rows [y0, y0 + h); groups of `group_size` samples
  uint32_t n_strips;
  uint32_t strip_x0[MAX_CR2_STRIPS];
  uint32_t strip_w[MAX_CR2_STRIPS];
  uint32_t strip_y0[MAX_CR2_STRIPS];
  uint32_t strip_h[MAX_CR2_STRIPS];
  uint64_t strip_first_sample[MAX_CR2_STRIPS + 1]; // prefix: stream sample index
  uint32_t job; // owning job (status / consumed are reported per job)
  // kind 2 (NikonDecompressor): plain MSB bit stream (BitStreamerMSB)
  uint8_t raw;       // no FF00 un-stuffing, no markers, 8-byte over-read budget
  uint8_t start_bit; // the first symbol starts this many bits into in_offset
  uint8_t las;       // table 0 holds "lossy after split" values
  uint8_t pair;      // Hasselblad: pair-coded symbols on an MSB32 stream
  uint8_t no_vertical; // rows do not inherit predictors from the row above
  uint64_t raw_limit; // != 0: last bit offset at which a symbol may start
};

int build_ljpeg_stream(const rsx_ljpeg_desc& d, const rsx_image& img,
                       StreamGeom* g);
int build_cr2_stream(const rsx_cr2_desc& d, const rsx_image& img,
                     StreamGeom* g);

// ------------------------------------------------------------------------
// Context / plan
// ------------------------------------------------------------------------
// Owning handle of a device allocation (blocks are recycled through a small
// size-keyed cache, rsx_host.cpp).  Released on destruction, so every early return of
// the plan builders gives its memory back.
struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : ptr(o.ptr), bytes(o.bytes) {
    o.ptr = nullptr;
    o.bytes = 0;
  }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
      release();
      ptr = o.ptr;
      bytes = o.bytes;
      o.ptr = nullptr;
      o.bytes = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  int ensure(size_t n); // grow-only; returns RSX_OK / RSX_ERR_NOMEM
  void release();
};

// ------------------------------------------------------------------------
// Helper threads of a context's host-pointer calls (the uploader of a banded unpack call,
// the bands of a split DNG call): a few PERSISTENT workers that sleep on a condition
// variable, started on first use and kept -- until round 5 every such call created its
// own std::thread and the caller spun on an atomic with yield().  Tasks never wait for
// other tasks, so a full pool only queues.
class HelperPool {
public:
  struct Task {
    std::function<void()> fn;
    bool done = false;
    bool failed = false; // fn threw
  };
  typedef std::shared_ptr<Task> Handle;
  ~HelperPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      quit_ = true;
    }
    cv_work_.notify_all();
    for (std::thread& t : threads_)
      if (t.joinable())
        t.join();
  }
  // nullptr: no worker could be had -- the caller runs fn itself
  Handle submit(std::function<void()> fn) {
    Handle h = std::make_shared<Task>();
    h->fn = std::move(fn);
    {
      std::lock_guard<std::mutex> g(m_);
      if (idle_ <= int(queue_.size()) && int(threads_.size()) < MAX_THREADS) {
        try {
          threads_.emplace_back([this]() { work(); });
        } catch (...) {
          if (threads_.empty())
            return nullptr;
        }
      }
      queue_.push_back(h);
    }
    cv_work_.notify_one();
    return h;
  }
  void wait(const Handle& h) {
    if (!h)
      return;
    std::unique_lock<std::mutex> g(m_);
    cv_done_.wait(g, [&]() { return h->done; });
  }

private:
  static constexpr int MAX_THREADS = 16;
  void work() {
    std::unique_lock<std::mutex> g(m_);
    for (;;) {
      ++idle_;
      cv_work_.wait(g, [&]() { return quit_ || !queue_.empty(); });
      --idle_;
      if (queue_.empty())
        return; // (quit)
      Handle h = queue_.front();
      queue_.pop_front();
      g.unlock();
      // (a task that throws -- std::bad_alloc from a band's vectors -- must not take the
      // process down from a pool thread, and its waiter must still wake up: the task's own
      // result fields say "not done", which its caller turns into a status)
      try {
        h->fn();
      } catch (...) {
        h->failed = true;
      }
      g.lock();
      h->done = true;
      cv_done_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  std::deque<Handle> queue_;
  std::vector<std::thread> threads_;
  int idle_ = 0;
  bool quit_ = false;
};

} // namespace rsx

struct rsx_ctx {
  int device = 0;
  hipStream_t stream = nullptr; // default stream of the plan API (rsx_plan_run(.., NULL))
  // Plan bookkeeping takes this for a moment (device selection, uploads); the
  // host-pointer calls do NOT hold it while they run: each takes a lane of its own.
  std::recursive_mutex mu;
  std::mutex err_mu;
  std::string last_error;
  std::atomic<uint64_t> host_calls{0}; // host-pointer entry points served
  std::atomic<uint64_t> chunked_calls{0}; // ... of which ran one large stream in chunks
  bool host_overlap = true;            // large unpack-family host calls run in row bands
  std::mutex upload_mu, download_mu;   // LJPEG-family host calls: one copy per direction at a time
  // The single-pass LJPEG kernel takes a workgroup's place in its stream's order from its
  // block index (rsx_ljpeg_fast.hip): safe while the dispatcher starts one grid's workgroups
  // in order, but two such grids on two HIP streams could each fill the slots the other's
  // next workgroup needs.  Its launches therefore run one after the other per context: each
  // waits for the event recorded behind the last one (the kernels around it still overlap).
  rsx::HelperPool helpers;             // persistent helper threads of the host-pointer calls
  hipEvent_t null_ev = nullptr;        // rsx_plan_run(stream == NULL): the context's stream behind the null stream's work so far
  std::mutex fast_mu;
  hipEvent_t fast_ev = nullptr;
  hipStream_t fast_ev_stream = nullptr;
  bool fast_ev_valid = false;
  // Staging of one host-pointer call: device buffers + a stream.  Lanes are pooled, so
  // calls from different threads (rstest-style file loops, DNG tile threads of an
  // unbatched build) stage and decode side by side instead of queueing on one mutex.
  struct HostLane {
    rsx::DeviceBuffer d_in, d_out;
    hipStream_t stream = nullptr;
    // the plan of the lane's last LJPEG-family call and what it was made from: a caller
    // that decodes the same layout again (a burst, a benchmark loop, the tiles of one
    // camera's files) skips the plan's construction -- tables, block lists, a dozen uploads
    struct rsx_plan* cached_plan = nullptr;
    std::vector<uint8_t> cached_key;
    // page-locked staging of the lane's downloads (rsx_api.hip, download_rects): what does not
    // lie on the 16-byte grid in the caller's memory goes through here, in two halves that take
    // turns (the copy of one chunk under the host's scatter of the one before)
    uint8_t* h_pin = nullptr;
    size_t h_pin_bytes = 0;
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    // upload stream + one event per band of the overlapped host path (rsx_api.hip, unpack_host)
    hipStream_t stream_up = nullptr;
    std::vector<hipEvent_t> ev_up;
    bool ensure_overlap(int bands) {
      if (!stream_up && hipStreamCreateWithFlags(&stream_up, hipStreamNonBlocking) != hipSuccess)
        return false;
      while (int(ev_up.size()) < bands) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
          return false;
        ev_up.push_back(e);
      }
      return true;
    }
  };
  static constexpr int RSX_MAX_LANES = 16;
  std::mutex lanes_mu;
  std::condition_variable lanes_cv;
  std::vector<std::unique_ptr<HostLane>> lanes_all;
  std::vector<HostLane*> lanes_free;
  HostLane* acquire_lane(const std::vector<uint8_t>* want_key = nullptr);
  void release_lane(HostLane* l);
};

#define RSX_HIP_CHECK(ctx, expr)                                               \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      if (ctx) {                                                               \
        std::lock_guard<std::mutex> _g((ctx)->err_mu);                         \
        (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e); \
      }                                                                        \
      return RSX_ERR_DEVICE;                                                   \
    }                                                                          \
  } while (0)
