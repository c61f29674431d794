// rsx_api.hip -- the C-ABI entry points declared in include/rsx.h.
//
// Host glue only: validation, staging, plan bookkeeping, launches.  There is
// no CPU decode path anywhere in this library: without a GPU, context creation
// fails with RSX_ERR_DEVICE.
#include "rsx_device.h"
#include "rsx_internal.h"
#include "rsx_ljpeg.h"
#include "rsx_ljpeg_dev.h"
#include "rsx_samsung_v2.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstring>
#include <memory>

using namespace rsx;

// ---------------------------------------------------------------------------
// Plan
// ---------------------------------------------------------------------------
namespace {

enum PlanKind { PLAN_UNPACK = 0, PLAN_LJPEG = 1, PLAN_SRAW = 2, PLAN_SV2 = 3 };

struct UnpackLaunch {
  int mode = UNPACK_MODE_PACKED;
  int order = 0;
  int n_jobs = 0;
  uint32_t total_blocks = 0;
  DeviceBuffer d_jobs, d_block_start;
  std::vector<UnpackJobDev> jobs; // host copy (alignment flags patched per base)
  uintptr_t last_out_base = ~uintptr_t(0);
};

struct EventPair {
  hipEvent_t start = nullptr, stop = nullptr;
};

} // namespace

struct rsx_plan {
  rsx_ctx* ctx = nullptr;
  PlanKind kind = PLAN_UNPACK;
  int n_jobs = 0;
  std::vector<int32_t> job_status; // host-side validation result per job
  std::vector<UnpackLaunch> unpack;
  std::unique_ptr<LJpegPlan, LJpegPlanDeleter> ljpeg;
  Sv2Plan* sv2 = nullptr; // PLAN_SV2
  // PLAN_SRAW
  DeviceBuffer d_sraw_jobs, d_sraw_starts;
  int n_sraw = 0;
  uint32_t sraw_blocks = 0;
  bool sraw_versions[3] = {false, false, false};
  hipStream_t last_stream = nullptr;
  bool ran = false;
  // dominant-kernel timing
  bool timing = false;
  std::vector<EventPair> events;
  size_t events_used = 0;
  // LJPEG plans: an event after every kernel of a timed run; the totals per kernel
  // name are folded in before the events are reused
  std::unique_ptr<KernelTimer> ktimer;
  bool ktimer_pending = false;
  std::vector<std::pair<const char*, double>> ktotals;
  int kruns = 0;
};

namespace {

int upload(rsx_ctx* ctx, DeviceBuffer& buf, const void* src, size_t bytes) {
  if (int st = buf.ensure(bytes ? bytes : 16))
    return st;
  if (bytes)
    RSX_HIP_CHECK(ctx, hipMemcpy(buf.ptr, src, bytes, hipMemcpyHostToDevice));
  return RSX_OK;
}

// The pool is created by rsx_plan_set_timing(); once it is exhausted further
// launches simply go untimed (event creation is far too slow for a hot path).
EventPair* next_events(rsx_plan* p) {
  if (p->events_used == p->events.size())
    return nullptr;
  return &p->events[p->events_used++];
}

// add the durations of the last timed LJPEG run to the per-kernel totals
int fold_kernel_timer(rsx_plan* p) {
  if (!p->ktimer || !p->ktimer_pending)
    return RSX_OK;
  rsx_ctx* ctx = p->ctx;
  KernelTimer& t = *p->ktimer;
  p->ktimer_pending = false;
  if (t.n == 0)
    return RSX_OK;
  RSX_HIP_CHECK(ctx, hipEventSynchronize(t.ev[t.n]));
  for (int i = 0; i < t.n; ++i) {
    float ms = 0;
    RSX_HIP_CHECK(ctx, hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
    bool found = false;
    for (auto& kv : p->ktotals)
      if (kv.first == t.name[i] || std::strcmp(kv.first, t.name[i]) == 0) {
        kv.second += ms;
        found = true;
        break;
      }
    if (!found)
      p->ktotals.emplace_back(t.name[i], double(ms));
  }
  ++p->kruns;
  return RSX_OK;
}

int flatten_unpack_job(const rsx_unpack_job& j, UnpackJobDev* out, int* order) {
  const rsx_unpack_desc& d = j.desc;
  if (int st = validate_unpack(d, j.img, size_t(j.in_bytes)))
    return st;
  if (j.img.pitch_bytes % 2 != 0)
    return RSX_ERR_INVALID_ARG;
  UnpackJobDev u{};
  u.in_offset = j.in_offset;
  u.stream_bytes = uint64_t(d.crop_h) * uint64_t(d.input_pitch_bytes);
  u.in_pitch = uint32_t(d.input_pitch_bytes);
  u.out_pitch = j.img.pitch_bytes;
  // h = min(h + oy, dim.y); rows [oy, h)  (UncompressedDecompressor.cpp:209-210)
  const int64_t rows_avail = int64_t(j.img.dim_y) - d.crop_y;
  u.n_rows = uint32_t(std::min<int64_t>(d.crop_h, rows_avail));
  u.cols = uint32_t(d.crop_w) * uint32_t(j.img.cpp);
  u.bps = uint32_t(d.bits_per_pixel);
  uint64_t out_off = j.img_offset + uint64_t(d.crop_y) * j.img.pitch_bytes;
  // Only the 16-bit LSB copyPixels path honours offset.x (:257-264); the
  // packed paths write from column 0 (:196) -- replicated.
  if (d.bit_order == RSX_ORDER_LSB && d.bits_per_pixel == 16)
    out_off += uint64_t(d.crop_x) * uint64_t(j.img.cpp) * 2;
  u.out_offset = out_off;
  unpack_blocks_for(&u);
  u.out_aligned = 0; // resolved at run time (needs the output base pointer)
  *out = u;
  *order = d.bit_order;
  return RSX_OK;
}

} // namespace

// ---------------------------------------------------------------------------
// Misc
// ---------------------------------------------------------------------------
// The lane pool is capped (RSX_MAX_LANES: every lane owns a stream, staging for a whole
// image and a cached plan with its scratch): a caller beyond the cap waits for a lane to
// come back instead of growing the pool with the thread count of the host program.
rsx_ctx::HostLane* rsx_ctx::acquire_lane(const std::vector<uint8_t>* want_key) {
  std::unique_lock<std::mutex> g(lanes_mu);
  while (true) {
    if (!lanes_free.empty()) {
      // (a lane whose cached plan was made from the same jobs, if there is one: the bands of
      // a split DNG call come back in any order)
      size_t pick = lanes_free.size() - 1;
      if (want_key)
        for (size_t k = 0; k < lanes_free.size(); ++k)
          if (lanes_free[k]->cached_plan && lanes_free[k]->cached_key == *want_key) {
            pick = k;
            break;
          }
      HostLane* l = lanes_free[pick];
      lanes_free.erase(lanes_free.begin() + long(pick));
      return l;
    }
    if (lanes_all.size() < size_t(RSX_MAX_LANES))
      break;
    lanes_cv.wait(g);
  }
  auto l = std::make_unique<HostLane>();
  if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess)
    return nullptr;
  lanes_all.push_back(std::move(l));
  return lanes_all.back().get();
}

void rsx_ctx::release_lane(HostLane* l) {
  {
    std::lock_guard<std::mutex> g(lanes_mu);
    lanes_free.push_back(l);
  }
  lanes_cv.notify_one();
}

namespace {
struct LaneGuard {
  rsx_ctx* ctx;
  rsx_ctx::HostLane* lane;
  explicit LaneGuard(rsx_ctx* c, const std::vector<uint8_t>* want_key = nullptr)
      : ctx(c), lane(c->acquire_lane(want_key)) {}
  LaneGuard(const LaneGuard&) = delete;
  ~LaneGuard() {
    if (lane)
      ctx->release_lane(lane);
  }
};
} // namespace

static void set_error(rsx_ctx* ctx, const std::string& msg) {
  std::lock_guard<std::mutex> g(ctx->err_mu);
  ctx->last_error = msg;
}

// ---------------------------------------------------------------------------------------
// Downloads into the CALLER's memory (round 6; DESIGN 7 "the undelivered sixteen bytes").
//
// The caller's image is pageable.  A device-to-host copy into it is carried out by the
// runtime on the caller's pages (pinned for the copy); for a rectangle whose start, pitch or
// width is off the 16-byte grid that is a byte-granular copy kernel (16 x 16 lanes, a byte
// each), and ONE such copy left one 16-lane row segment of one workgroup undelivered in a
// few images of several hundred (round 5: 3-sample pixels, tiles of unequal heights; status
// OK, the caller's fill showing through; never reproduced outside the library, never on a
// copy that lies on the grid).  The rule since round 6, at EVERY download of the library:
//   * what lies on the 16-byte grid in the caller's memory AND on the device (start, pitch,
//     width) is copied straight into the image -- the path every frame of every benchmark
//     and test has taken since round 1;
//   * everything else goes through page-locked staging of the lane (hipHostMalloc): the
//     device side of such a copy is widened to the grid, so the runtime only ever sees
//     aligned copies into memory it pinned itself, and the host moves the bytes the caller
//     owns into the image.  A rectangle that merely starts or ends off the grid (an odd
//     width, a tile in the middle of a 3-sample image) is split: its aligned body goes
//     straight, the < 16 bytes a row on either side are staged as 16-byte columns.
// Nothing is written outside the rectangles (the reference's contract: row padding and other
// tiles' pixels are never touched, LJpegDecompressor.cpp:264-268,
// AbstractDngDecompressor.cpp:112-131).
// ---------------------------------------------------------------------------------------
namespace {

struct DownRect {
  uint8_t* host;      // first byte of the rectangle in the caller's image
  size_t host_pitch;
  const uint8_t* dev; // first byte of the rectangle on the device
  size_t dev_pitch;
  size_t bytes, rows; // width in bytes, rows
};

constexpr size_t PIN_HALF = size_t(8) << 20; // a half of the lane's staging

// a piece that goes through the staging: rows of `width` device bytes from `dev` (every
// dev_pitch; on the 16-byte grid) or, `linear`, the contiguous device range that holds them;
// of row y the host takes `take` bytes from offset `skip` (+ y * dev_pitch when linear)
struct StagedPiece {
  const uint8_t* dev;
  size_t dev_pitch, width, rows;
  bool linear;
  uint8_t* host;
  size_t host_pitch, skip, take;
};

int ensure_pin(rsx_ctx* ctx, rsx_ctx::HostLane* L) {
  if (!L->h_pin) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 2 * PIN_HALF, hipHostMallocDefault) != hipSuccess || !p) {
      (void)hipGetLastError();
      return RSX_ERR_NOMEM;
    }
    L->h_pin = static_cast<uint8_t*>(p);
    L->h_pin_bytes = 2 * PIN_HALF;
  }
  for (hipEvent_t& e : L->ev_pin)
    if (!e)
      RSX_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return RSX_OK;
}

inline bool on_grid(uintptr_t v) { return (v & 15u) == 0; }

// Queues the copies of `n` rectangles on `s` and returns when every byte is in the caller's
// memory (the stream is synchronised).  The device rows must be final on `s`.
int download_rects(rsx_ctx* ctx, rsx_ctx::HostLane* L, hipStream_t s, const DownRect* rects,
                   size_t n) {
  std::vector<StagedPiece> staged;
  hipError_t err = hipSuccess;
  auto direct = [&](uint8_t* host, size_t hp, const uint8_t* dev, size_t dp, size_t bytes,
                    size_t rows) {
    if (err != hipSuccess || bytes == 0 || rows == 0)
      return;
    if (rows == 1 || (bytes == hp && bytes == dp))
      err = hipMemcpyAsync(host, dev, rows == 1 ? bytes : bytes * rows, hipMemcpyDeviceToHost, s);
    else
      err = hipMemcpy2DAsync(host, hp, dev, dp, bytes, rows, hipMemcpyDeviceToHost, s);
  };
  // Several rectangles in one call (tiles of unequal heights side by side, a tile set with a
  // failed tile: what does not merge into one rectangle) share the pages of their rows.  Copied
  // straight, each would have the runtime pin its own, overlapping range of the caller's pages
  // -- the one constellation in which bytes were ever lost (profiles/r06/host_path_defect.md).
  // They all go through the staging; ONE rectangle's body goes straight.
  size_t n_live = 0;
  for (size_t i = 0; i < n; ++i)
    n_live += rects[i].bytes != 0 && rects[i].rows != 0;
  const bool all_staged = n_live >= 2;
  for (size_t i = 0; i < n; ++i) {
    const DownRect& r = rects[i];
    if (r.bytes == 0 || r.rows == 0)
      continue;
    const uintptr_t ha = reinterpret_cast<uintptr_t>(r.host), da = reinterpret_cast<uintptr_t>(r.dev);
    const bool pitches = (r.rows == 1) || (on_grid(r.host_pitch) && on_grid(r.dev_pitch));
    if (all_staged) {
      const size_t mis = da & 15u;
      if (on_grid(r.dev_pitch) || r.rows == 1)
        staged.push_back({r.dev - mis, r.dev_pitch, (mis + r.bytes + 15) & ~size_t(15), r.rows, false,
                          r.host, r.host_pitch, mis, r.bytes});
      else
        staged.push_back({r.dev, r.dev_pitch, r.bytes, r.rows, true, r.host, r.host_pitch, 0, r.bytes});
      continue;
    }
    if (pitches && on_grid(ha) && on_grid(da) && on_grid(r.bytes)) {
      direct(r.host, r.host_pitch, r.dev, r.dev_pitch, r.bytes, r.rows);
      continue;
    }
    if (pitches && (ha & 15u) == (da & 15u)) {
      // [head < 16][body on the grid][tail < 16]
      const size_t mis = ha & 15u;
      const size_t head = std::min(r.bytes, (16u - mis) & 15u);
      const size_t body = (r.bytes - head) & ~size_t(15);
      const size_t tail = r.bytes - head - body;
      if (head)
        staged.push_back({r.dev - mis, r.dev_pitch, 16, r.rows, false, r.host, r.host_pitch, mis, head});
      direct(r.host + head, r.host_pitch, r.dev + head, r.dev_pitch, body, r.rows);
      if (tail)
        staged.push_back({r.dev + head + body, r.dev_pitch, 16, r.rows, false,
                          r.host + head + body, r.host_pitch, 0, tail});
      continue;
    }
    // the host and the device disagree about the grid (a compact device rectangle for a
    // cropped host one), or a pitch is off it: everything through the staging
    const size_t mis = da & 15u;
    if (on_grid(r.dev_pitch) || r.rows == 1)
      staged.push_back({r.dev - mis, r.dev_pitch, (mis + r.bytes + 15) & ~size_t(15), r.rows, false,
                        r.host, r.host_pitch, mis, r.bytes});
    else
      staged.push_back({r.dev, r.dev_pitch, r.bytes, r.rows, true, r.host, r.host_pitch, 0, r.bytes});
  }
  if (err == hipSuccess && !staged.empty()) {
    if (int e = ensure_pin(ctx, L)) {
      (void)hipStreamSynchronize(s);
      return e;
    }
    // chunks = (piece, row range) that fit a half; copy of chunk c + 1 under the scatter of chunk c
    struct Chunk {
      size_t piece, y0, y1, lin_lo; // lin_lo: first device byte of a linear chunk, relative to piece.dev
    };
    std::vector<Chunk> chunks;
    for (size_t k = 0; k < staged.size(); ++k) {
      const StagedPiece& p = staged[k];
      const size_t per_row = p.linear ? p.dev_pitch : p.width;
      const size_t rows_per = std::max<size_t>(1, (PIN_HALF - 64) / std::max<size_t>(per_row, 1));
      for (size_t y = 0; y < p.rows; y += rows_per)
        chunks.push_back({k, y, std::min(p.rows, y + rows_per), 0});
    }
    auto issue = [&](size_t c) {
      Chunk& ch = chunks[c];
      const StagedPiece& p = staged[ch.piece];
      uint8_t* half = L->h_pin + (c & 1u) * PIN_HALF;
      if (!p.linear) {
        const size_t rows = ch.y1 - ch.y0;
        if (p.width > PIN_HALF - 64) { // (a single row wider than a half: cannot happen for images of < 8 MB a row)
          err = hipErrorInvalidValue;
          return;
        }
        err = rows == 1 || p.width == p.dev_pitch
                  ? hipMemcpyAsync(half, p.dev + ch.y0 * p.dev_pitch, p.width * rows, hipMemcpyDeviceToHost, s)
                  : hipMemcpy2DAsync(half, p.width, p.dev + ch.y0 * p.dev_pitch, p.dev_pitch, p.width,
                                     rows, hipMemcpyDeviceToHost, s);
      } else {
        // the device bytes of rows y0 .. y1 - 1, from the 16-byte boundary in front of them to
        // the one behind (the lane's device buffers start on the grid and end with slack)
        const uintptr_t d0 = reinterpret_cast<uintptr_t>(p.dev) + ch.y0 * p.dev_pitch;
        const uintptr_t d1 = reinterpret_cast<uintptr_t>(p.dev) + (ch.y1 - 1) * p.dev_pitch + p.take;
        const uintptr_t lo = d0 & ~uintptr_t(15), hi = (d1 + 15) & ~uintptr_t(15);
        ch.lin_lo = size_t(d0 - lo);
        if (hi - lo > PIN_HALF) {
          err = hipErrorInvalidValue;
          return;
        }
        err = hipMemcpyAsync(half, reinterpret_cast<const void*>(lo), size_t(hi - lo),
                             hipMemcpyDeviceToHost, s);
      }
      if (err == hipSuccess)
        err = hipEventRecord(L->ev_pin[c & 1u], s);
    };
    for (size_t c = 0; c < chunks.size() && c < 2 && err == hipSuccess; ++c)
      issue(c);
    for (size_t c = 0; c < chunks.size() && err == hipSuccess; ++c) {
      err = hipEventSynchronize(L->ev_pin[c & 1u]);
      if (err != hipSuccess)
        break;
      const Chunk& ch = chunks[c];
      const StagedPiece& p = staged[ch.piece];
      const uint8_t* half = L->h_pin + (c & 1u) * PIN_HALF;
      for (size_t y = ch.y0; y < ch.y1; ++y) {
        const uint8_t* src = p.linear ? half + ch.lin_lo + (y - ch.y0) * p.dev_pitch
                                      : half + (y - ch.y0) * p.width + p.skip;
        std::memcpy(p.host + y * p.host_pitch, src, p.take);
      }
      if (c + 2 < chunks.size())
        issue(c + 2);
    }
  }
  // (on every way out: the caller's image may be freed the moment the call returns)
  const hipError_t es = hipStreamSynchronize(s);
  if (err == hipSuccess)
    err = es;
  if (err != hipSuccess) {
    set_error(ctx, std::string("download: ") + hipGetErrorString(err));
    return RSX_ERR_DEVICE;
  }
  return RSX_OK;
}

} // namespace

extern "C" int rsx_abi_version(void) { return RSX_ABI_VERSION; }

extern "C" const char* rsx_status_string(int status) {
  switch (status) {
  case RSX_OK: return "RSX_OK";
  case RSX_ERR_INVALID_ARG: return "RSX_ERR_INVALID_ARG";
  case RSX_ERR_IO: return "RSX_ERR_IO";
  case RSX_ERR_BAD_HUFFMAN_CODE: return "RSX_ERR_BAD_HUFFMAN_CODE";
  case RSX_ERR_RESTART_MARKER: return "RSX_ERR_RESTART_MARKER";
  case RSX_ERR_INPUT_OVERFLOW: return "RSX_ERR_INPUT_OVERFLOW";
  case RSX_ERR_DEVICE: return "RSX_ERR_DEVICE";
  case RSX_ERR_UNSUPPORTED: return "RSX_ERR_UNSUPPORTED";
  case RSX_ERR_NOMEM: return "RSX_ERR_NOMEM";
  case RSX_ERR_TILE_ERRORS: return "RSX_ERR_TILE_ERRORS";
  case RSX_ERR_VALUE_RANGE: return "RSX_ERR_VALUE_RANGE";
  default: return "RSX_ERR_UNKNOWN";
  }
}

extern "C" int rsx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

extern "C" int rsx_ctx_create(int device, rsx_ctx** out_ctx) {
  if (!out_ctx)
    return RSX_ERR_INVALID_ARG;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    return RSX_ERR_DEVICE; // no GPU: fail loudly, there is no CPU fallback
  if (hipSetDevice(device) != hipSuccess)
    return RSX_ERR_DEVICE;
  auto* ctx = new rsx_ctx();
  ctx->device = device;
  // (read once per context, not per call: RSX_HOST_NO_OVERLAP=1 keeps large host-pointer
  // calls of the unpack family in one piece -- what the tests compare the banded path with)
  ctx->host_overlap = getenv("RSX_HOST_NO_OVERLAP") == nullptr;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return RSX_ERR_DEVICE;
  }
  *out_ctx = ctx;
  return RSX_OK;
}

extern "C" void rsx_ctx_destroy(rsx_ctx* ctx) {
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  for (auto& l : ctx->lanes_all) {
    // the cached plan first: its destructor synchronises the stream it last ran on,
    // which is this lane's
    if (l->stream)
      (void)hipStreamSynchronize(l->stream);
    if (l->cached_plan)
      rsx_plan_destroy(l->cached_plan);
    l->cached_plan = nullptr;
    if (l->stream)
      (void)hipStreamDestroy(l->stream);
    for (hipEvent_t e : l->ev_up)
      (void)hipEventDestroy(e);
    l->ev_up.clear();
    if (l->stream_up)
      (void)hipStreamDestroy(l->stream_up);
    l->d_in.release();
    l->d_out.release();
    if (l->h_pin)
      (void)hipHostFree(l->h_pin);
    l->h_pin = nullptr;
    for (hipEvent_t& e : l->ev_pin) {
      if (e)
        (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  if (ctx->fast_ev)
    (void)hipEventDestroy(ctx->fast_ev);
  if (ctx->null_ev)
    (void)hipEventDestroy(ctx->null_ev);
  delete ctx;
}

// (a copy made under the lock, per calling thread: host-pointer calls from several
// threads may be writing the context's string at the same time)
extern "C" const char* rsx_ctx_last_error(const rsx_ctx* ctx) {
  if (!ctx)
    return "";
  static thread_local std::string copy;
  {
    std::lock_guard<std::mutex> g(const_cast<rsx_ctx*>(ctx)->err_mu);
    copy = ctx->last_error;
  }
  return copy.c_str();
}

extern "C" uint64_t rsx_ctx_host_calls(const rsx_ctx* ctx) {
  return ctx ? ctx->host_calls.load() : 0;
}
extern "C" uint64_t rsx_ctx_chunked_calls(const rsx_ctx* ctx) {
  return ctx ? ctx->chunked_calls.load() : 0;
}

// Page-locked host memory (rsx.h: optional; replaces nothing in the reference, it changes
// where RawImageData::createData, RawImage.cpp:68-100, and the file Buffer get their bytes).
extern "C" int rsx_host_alloc(rsx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out || bytes == 0)
    return RSX_ERR_INVALID_ARG;
  *out = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return RSX_ERR_DEVICE;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
    (void)hipGetLastError();
    return RSX_ERR_NOMEM;
  }
  *out = p;
  return RSX_OK;
}

extern "C" int rsx_host_free(rsx_ctx* ctx, void* p) {
  if (!ctx)
    return RSX_ERR_INVALID_ARG;
  if (!p)
    return RSX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return RSX_ERR_DEVICE;
  if (hipHostFree(p) != hipSuccess) {
    (void)hipGetLastError();
    return RSX_ERR_INVALID_ARG;
  }
  return RSX_OK;
}

extern "C" int rsx_host_register(rsx_ctx* ctx, void* p, size_t bytes) {
  if (!ctx || !p || bytes == 0)
    return RSX_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return RSX_ERR_DEVICE;
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e == hipErrorHostMemoryAlreadyRegistered ? RSX_OK : RSX_ERR_NOMEM;
  }
  return RSX_OK;
}

extern "C" int rsx_host_unregister(rsx_ctx* ctx, void* p) {
  if (!ctx || !p)
    return RSX_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess)
    return RSX_ERR_DEVICE;
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return RSX_ERR_INVALID_ARG;
  }
  return RSX_OK;
}

extern "C" int rsx_unpack_validate(const rsx_unpack_desc* d, const rsx_image* img,
                                   size_t in_bytes) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_unpack(*d, *img, in_bytes);
}

extern "C" int rsx_ljpeg_validate(const rsx_ljpeg_desc* d, const rsx_image* img,
                                  size_t in_bytes) {
  (void)in_bytes;
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_ljpeg(*d, *img);
}

extern "C" int rsx_cr2_validate(const rsx_cr2_desc* d, const rsx_image* img,
                                size_t in_bytes) {
  (void)in_bytes;
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_cr2(*d, *img);
}

// ---------------------------------------------------------------------------
// Plans
// ---------------------------------------------------------------------------
extern "C" int rsx_unpack_plan_create(rsx_ctx* ctx, int n_jobs,
                                      const rsx_unpack_job* jobs,
                                      rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_UNPACK;
  plan->n_jobs = n_jobs;
  plan->job_status.assign(n_jobs, RSX_OK);
  std::vector<UnpackJobDev> per_order[4];
  for (int i = 0; i < n_jobs; ++i) {
    UnpackJobDev u;
    int order = 0;
    const int st = flatten_unpack_job(jobs[i], &u, &order);
    plan->job_status[i] = st;
    if (st == RSX_OK && u.n_rows > 0)
      per_order[order].push_back(u);
  }
  for (int order = 0; order < 4; ++order) {
    auto& v = per_order[order];
    if (v.empty())
      continue;
    plan->unpack.emplace_back();
    UnpackLaunch& L = plan->unpack.back();
    L.order = order;
    L.n_jobs = int(v.size());
    std::vector<uint32_t> starts(v.size() + 1, 0);
    for (size_t k = 0; k < v.size(); ++k)
      starts[k + 1] = starts[k] + v[k].n_rows * v[k].segs_per_row;
    L.total_blocks = starts.back();
    L.jobs = v;
    if (int st = upload(ctx, L.d_jobs, v.data(), v.size() * sizeof(UnpackJobDev)))
      return st;
    if (int st = upload(ctx, L.d_block_start, starts.data(),
                        starts.size() * sizeof(uint32_t)))
      return st;
  }
  *out_plan = plan.release();
  return RSX_OK;
}

// F32 images: readUncompressedRaw's floating-point branches (:212-245)
extern "C" int rsx_unpack_f32_validate(const rsx_unpack_desc* d, const rsx_image* img,
                                       size_t in_bytes) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_unpack_f32(*d, *img, in_bytes);
}

extern "C" int rsx_unpack_f32_plan_create(rsx_ctx* ctx, int n_jobs,
                                          const rsx_unpack_job* jobs,
                                          rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_UNPACK;
  plan->n_jobs = n_jobs;
  plan->job_status.assign(n_jobs, RSX_OK);
  struct Class {
    int order, bps;
    std::vector<UnpackJobDev> v;
  };
  Class classes[5] = {{RSX_ORDER_LSB, 32, {}}, {RSX_ORDER_LSB, 16, {}},
                      {RSX_ORDER_LSB, 24, {}}, {RSX_ORDER_MSB, 16, {}},
                      {RSX_ORDER_MSB, 24, {}}};
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_unpack_job& j = jobs[i];
    const rsx_unpack_desc& d = j.desc;
    int st = validate_unpack_f32(d, j.img, size_t(j.in_bytes));
    if (st == RSX_OK && j.img.pitch_bytes % 4 != 0)
      st = RSX_ERR_INVALID_ARG;
    plan->job_status[i] = st;
    if (st != RSX_OK)
      continue;
    UnpackJobDev u{};
    u.in_offset = j.in_offset;
    u.stream_bytes = uint64_t(d.crop_h) * uint64_t(d.input_pitch_bytes);
    u.in_pitch = uint32_t(d.input_pitch_bytes);
    u.out_pitch = j.img.pitch_bytes;
    const int64_t rows_avail = int64_t(j.img.dim_y) - d.crop_y; // :209-210
    u.n_rows = uint32_t(std::min<int64_t>(d.crop_h, rows_avail));
    u.cols = uint32_t(d.crop_w) * uint32_t(j.img.cpp);
    u.bps = uint32_t(d.bits_per_pixel);
    // copyPixels starts at out(y, offset.x * cpp) (:216-217), decodePackedFP
    // writes out(row, offset.x + col) (:181)
    const uint64_t x0 = d.bits_per_pixel == 32 ? uint64_t(d.crop_x) * j.img.cpp
                                               : uint64_t(d.crop_x);
    u.out_offset = j.img_offset + uint64_t(d.crop_y) * j.img.pitch_bytes + x0 * 4;
    unpack_fp_blocks_for(&u);
    if (u.n_rows == 0)
      continue;
    int cls = 0;
    if (d.bits_per_pixel != 32)
      cls = (d.bit_order == RSX_ORDER_MSB ? 3 : 1) + (d.bits_per_pixel == 24 ? 1 : 0);
    classes[cls].v.push_back(u);
  }
  for (Class& c : classes) {
    if (c.v.empty())
      continue;
    plan->unpack.emplace_back();
    UnpackLaunch& L = plan->unpack.back();
    L.mode = UNPACK_MODE_FP;
    L.order = (c.bps << 8) | c.order;
    L.n_jobs = int(c.v.size());
    std::vector<uint32_t> starts(c.v.size() + 1, 0);
    for (size_t k = 0; k < c.v.size(); ++k)
      starts[k + 1] = starts[k] + c.v[k].n_rows * c.v[k].segs_per_row;
    L.total_blocks = starts.back();
    L.jobs = c.v;
    if (int st = upload(ctx, L.d_jobs, c.v.data(), c.v.size() * sizeof(UnpackJobDev)))
      return st;
    if (int st = upload(ctx, L.d_block_start, starts.data(),
                        starts.size() * sizeof(uint32_t)))
      return st;
  }
  *out_plan = plan.release();
  return RSX_OK;
}

// decode8BitRaw<true> is the 8-bit packed walk over rows of w bytes;
// decode12BitRawUnpackedLeftAligned<e> the 16-bit LSB/MSB walk + ">> 4";
// decode12BitRawWithControl<e> has its own kernel.  All three write from pixel
// (0,0) and count `w` in samples of the uncropped array (cpp is not applied).
extern "C" int rsx_unpack_variant_validate(const rsx_unpack_variant_desc* d,
                                           const rsx_image* img, size_t in_bytes) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_unpack_variant(*d, *img, in_bytes);
}

extern "C" int rsx_unpack_variant_plan_create(rsx_ctx* ctx, int n_jobs,
                                              const rsx_unpack_variant_job* jobs,
                                              rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_UNPACK;
  plan->n_jobs = n_jobs;
  plan->job_status.assign(n_jobs, RSX_OK);
  // launch classes: [mode][order]
  struct Class {
    int mode, order;
    std::vector<UnpackJobDev> v;
  };
  Class classes[6] = {{UNPACK_MODE_PACKED, RSX_ORDER_LSB, {}},
                      {UNPACK_MODE_SHIFT, RSX_ORDER_LSB, {}},
                      {UNPACK_MODE_SHIFT, RSX_ORDER_MSB, {}},
                      {UNPACK_MODE_CONTROL, RSX_ORDER_LSB, {}},
                      {UNPACK_MODE_CONTROL, RSX_ORDER_MSB, {}},
                      {UNPACK_MODE_LUT8, RSX_ORDER_LSB, {}}};
  std::vector<uint16_t> luts; // tables of the LUT8 class, in job order
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_unpack_variant_job& j = jobs[i];
    int st = validate_unpack_variant(j.desc, j.img, size_t(j.in_bytes));
    if (st == RSX_OK && j.img.pitch_bytes % 2 != 0)
      st = RSX_ERR_INVALID_ARG;
    plan->job_status[i] = st;
    if (st != RSX_OK)
      continue;
    uint64_t bpl = 0;
    unpack_variant_bytes_per_line(j.desc, &bpl);
    UnpackJobDev u{};
    u.in_offset = j.in_offset;
    u.stream_bytes = bpl * uint64_t(j.desc.h);
    u.in_pitch = uint32_t(bpl);
    u.out_pitch = j.img.pitch_bytes;
    u.n_rows = uint32_t(j.desc.h);
    u.cols = uint32_t(j.desc.w);
    u.out_offset = j.img_offset;
    int cls = 0;
    switch (j.desc.variant) {
    case RSX_UNPACK_8BIT_RAW:
      u.bps = 8;
      unpack_blocks_for(&u);
      cls = 0;
      break;
    case RSX_UNPACK_8BIT_LOOKUP:
      u.bps = 8;
      u.post_shift = uint32_t(luts.size() / 256);
      luts.insert(luts.end(), j.desc.lut, j.desc.lut + 256);
      unpack_blocks_for(&u);
      cls = 5;
      break;
    case RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED:
      u.bps = 16;
      u.post_shift = 4;
      unpack_blocks_for(&u);
      cls = j.desc.big_endian ? 2 : 1;
      break;
    default:
      u.bps = j.desc.big_endian ? 1 : 0;
      unpack_control_blocks_for(&u);
      cls = j.desc.big_endian ? 4 : 3;
      break;
    }
    classes[cls].v.push_back(u);
  }
  for (Class& c : classes) {
    if (c.v.empty())
      continue;
    plan->unpack.emplace_back();
    UnpackLaunch& L = plan->unpack.back();
    L.mode = c.mode;
    L.order = c.order;
    L.n_jobs = int(c.v.size());
    std::vector<uint32_t> starts(c.v.size() + 1, 0);
    for (size_t k = 0; k < c.v.size(); ++k)
      starts[k + 1] = starts[k] + c.v[k].n_rows * c.v[k].segs_per_row;
    L.total_blocks = starts.back();
    L.jobs = c.v;
    // the LUT8 class keeps its tables right behind the job array
    std::vector<uint8_t> blob(c.v.size() * sizeof(UnpackJobDev) +
                              (c.mode == UNPACK_MODE_LUT8 ? luts.size() * 2 : 0));
    std::memcpy(blob.data(), c.v.data(), c.v.size() * sizeof(UnpackJobDev));
    if (c.mode == UNPACK_MODE_LUT8)
      std::memcpy(blob.data() + c.v.size() * sizeof(UnpackJobDev), luts.data(),
                  luts.size() * 2);
    if (int st = upload(ctx, L.d_jobs, blob.data(), blob.size()))
      return st;
    if (int st = upload(ctx, L.d_block_start, starts.data(),
                        starts.size() * sizeof(uint32_t)))
      return st;
  }
  *out_plan = plan.release();
  return RSX_OK;
}

// Whether 16-byte stores are legal depends on the run-time output base; the
// flag is re-resolved (and re-uploaded) only when the base pointer changes.
namespace {

int run_unpack(rsx_plan* p, const void* in_dev, void* out_dev, hipStream_t s) {
  rsx_ctx* ctx = p->ctx;
  for (UnpackLaunch& L : p->unpack) {
    const uintptr_t base = reinterpret_cast<uintptr_t>(out_dev);
    if (base != L.last_out_base) {
      bool changed = false;
      for (auto& u : L.jobs) {
        const uintptr_t a = base + u.out_offset;
        const uint32_t al = ((a & 15) == 0 && (u.out_pitch & 15) == 0) ? 1u : 0u;
        if (al != u.out_aligned) {
          u.out_aligned = al;
          changed = true;
        }
      }
      if (changed)
        RSX_HIP_CHECK(ctx, hipMemcpyAsync(L.d_jobs.ptr, L.jobs.data(),
                                          L.jobs.size() * sizeof(UnpackJobDev),
                                          hipMemcpyHostToDevice, s));
      L.last_out_base = base;
    }
    EventPair* ev = p->timing ? next_events(p) : nullptr;
    if (ev)
      RSX_HIP_CHECK(ctx, hipEventRecord(ev->start, s));
    RSX_HIP_CHECK(ctx, launch_unpack_mode(L.mode, L.order,
                                          static_cast<const UnpackJobDev*>(L.d_jobs.ptr),
                                          static_cast<const uint32_t*>(L.d_block_start.ptr),
                                          L.n_jobs, L.total_blocks, in_dev, out_dev, s));
    if (ev)
      RSX_HIP_CHECK(ctx, hipEventRecord(ev->stop, s));
  }
  return RSX_OK;
}

} // namespace

extern "C" int rsx_plan_run(rsx_plan* plan, const void* in_dev, void* out_dev,
                            void* stream) {
  if (!plan || !in_dev || !out_dev)
    return RSX_ERR_INVALID_ARG;
  rsx_ctx* ctx = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
  if (!stream) {
    // The convenience path: the context's own stream is hipStreamNonBlocking, i.e. NOT ordered behind
    // what the caller has queued on the null stream -- a fill of the output buffer, a device-to-device
    // copy of the input, a hipMemset (asynchronous with respect to the host on this runtime).  The run is
    // put behind the null stream's work so far; a caller who passes a stream orders it himself.
    if (!ctx->null_ev)
      RSX_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->null_ev, hipEventDisableTiming));
    RSX_HIP_CHECK(ctx, hipEventRecord(ctx->null_ev, nullptr));
    RSX_HIP_CHECK(ctx, hipStreamWaitEvent(s, ctx->null_ev, 0));
  }
  plan->last_stream = s;
  plan->ran = true;
  if (plan->kind == PLAN_UNPACK)
    return run_unpack(plan, in_dev, out_dev, s);
  EventPair* ev = plan->timing ? next_events(plan) : nullptr;
  if (plan->kind == PLAN_SRAW) {
    if (ev)
      RSX_HIP_CHECK(ctx, hipEventRecord(ev->start, s));
    RSX_HIP_CHECK(ctx, launch_sraw(static_cast<const SrawJobDev*>(plan->d_sraw_jobs.ptr),
                                   static_cast<const uint32_t*>(plan->d_sraw_starts.ptr),
                                   plan->n_sraw, plan->sraw_blocks, plan->sraw_versions,
                                   in_dev, out_dev, s));
    if (ev)
      RSX_HIP_CHECK(ctx, hipEventRecord(ev->stop, s));
    return RSX_OK;
  }
  (void)ev;
  const bool sv2 = plan->kind == PLAN_SV2;
  if (!plan->timing)
    return sv2 ? samsung_v2_plan_run(plan->sv2, in_dev, out_dev, s, nullptr)
               : ljpeg_plan_run(plan->ljpeg.get(), in_dev, out_dev, s, nullptr);
  if (int st = fold_kernel_timer(plan)) // (waits for the previous timed run)
    return st;
  if (!plan->ktimer)
    plan->ktimer = std::make_unique<KernelTimer>();
  plan->ktimer_pending = true;
  return sv2 ? samsung_v2_plan_run(plan->sv2, in_dev, out_dev, s, plan->ktimer.get())
             : ljpeg_plan_run(plan->ljpeg.get(), in_dev, out_dev, s, plan->ktimer.get());
}

extern "C" int rsx_plan_results(rsx_plan* plan, int32_t* job_status,
                                uint32_t* job_consumed) {
  if (!plan)
    return RSX_ERR_INVALID_ARG;
  rsx_ctx* ctx = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (plan->ran)
    RSX_HIP_CHECK(ctx, hipStreamSynchronize(plan->last_stream));
  int rc = RSX_OK;
  if (plan->kind == PLAN_UNPACK || plan->kind == PLAN_SRAW) {
    for (int i = 0; i < plan->n_jobs; ++i) {
      if (job_status)
        job_status[i] = plan->job_status[i];
      if (job_consumed)
        job_consumed[i] = 0;
      if (plan->job_status[i] != RSX_OK)
        rc = plan->job_status[i];
    }
    return rc;
  }
  if (plan->kind == PLAN_SV2) {
    if (job_consumed)
      for (int i = 0; i < plan->n_jobs; ++i)
        job_consumed[i] = 0;
    return samsung_v2_plan_results(plan->sv2, plan->last_stream, plan->ran, job_status);
  }
  return ljpeg_plan_results(plan->ljpeg.get(), plan->last_stream, plan->ran,
                            job_status, job_consumed);
}

extern "C" int rsx_plan_set_timing(rsx_plan* plan, int enable) {
  if (!plan)
    return RSX_ERR_INVALID_ARG;
  plan->timing = enable != 0;
  plan->events_used = 0;
  plan->ktotals.clear();
  plan->kruns = 0;
  plan->ktimer_pending = false;
  if (plan->kind == PLAN_LJPEG || plan->kind == PLAN_SV2)
    return RSX_OK; // its events are created on the first timed run
  if (plan->timing && plan->events.size() < 64) {
    // event creation is slow on ROCm: pre-create the pool outside timed regions
    std::lock_guard<std::recursive_mutex> lock(plan->ctx->mu);
    (void)hipSetDevice(plan->ctx->device);
    while (plan->events.size() < 64) {
      EventPair e;
      if (hipEventCreate(&e.start) != hipSuccess ||
          hipEventCreate(&e.stop) != hipSuccess)
        return RSX_ERR_DEVICE;
      plan->events.push_back(e);
    }
  }
  return RSX_OK;
}

extern "C" int rsx_plan_kernel_table(rsx_plan* plan, int cap, const char** names,
                                     double* avg_ms, int* n_kernels, int* n_runs) {
  if (!plan || (plan->kind != PLAN_LJPEG && plan->kind != PLAN_SV2) || !plan->timing)
    return RSX_ERR_INVALID_ARG;
  rsx_ctx* ctx = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int st = fold_kernel_timer(plan))
    return st;
  if (plan->kruns == 0)
    return RSX_ERR_INVALID_ARG;
  int n = 0;
  for (const auto& kv : plan->ktotals) {
    if (n < cap) {
      if (names)
        names[n] = kv.first;
      if (avg_ms)
        avg_ms[n] = kv.second / plan->kruns;
    }
    ++n;
  }
  if (n_kernels)
    *n_kernels = n;
  if (n_runs)
    *n_runs = plan->kruns;
  return RSX_OK;
}

extern "C" int rsx_plan_kernel_time(rsx_plan* plan, const char** kernel_name,
                                    double* avg_ms, int* n_launches) {
  if (!plan || !plan->timing)
    return RSX_ERR_INVALID_ARG;
  rsx_ctx* ctx = plan->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (plan->kind == PLAN_LJPEG || plan->kind == PLAN_SV2) {
    // the dominant kernel = the one with the largest share of the timed runs
    if (int st = fold_kernel_timer(plan))
      return st;
    if (plan->kruns == 0 || plan->ktotals.empty())
      return RSX_ERR_INVALID_ARG;
    const auto* best = &plan->ktotals[0];
    for (const auto& kv : plan->ktotals)
      if (kv.second > best->second)
        best = &kv;
    if (kernel_name)
      *kernel_name = best->first;
    if (avg_ms)
      *avg_ms = best->second / plan->kruns;
    if (n_launches)
      *n_launches = plan->kruns;
    plan->ktotals.clear();
    plan->kruns = 0;
    return RSX_OK;
  }
  if (plan->events_used == 0)
    return RSX_ERR_INVALID_ARG;
  double total = 0;
  for (size_t i = 0; i < plan->events_used; ++i) {
    RSX_HIP_CHECK(ctx, hipEventSynchronize(plan->events[i].stop));
    float ms = 0;
    RSX_HIP_CHECK(ctx, hipEventElapsedTime(&ms, plan->events[i].start,
                                           plan->events[i].stop));
    total += ms;
  }
  if (kernel_name)
    *kernel_name = plan->kind == PLAN_SRAW ? "sraw_kernel"
                   : (!plan->unpack.empty() &&
                      plan->unpack[0].mode == UNPACK_MODE_CONTROL)
                       ? "unpack_control_kernel"
                       : unpack_kernel_name();
  if (avg_ms)
    *avg_ms = total / double(plan->events_used);
  if (n_launches)
    *n_launches = int(plan->events_used);
  plan->events_used = 0;
  return RSX_OK;
}

extern "C" void rsx_plan_destroy(rsx_plan* plan) {
  if (!plan)
    return;
  rsx_ctx* ctx = plan->ctx;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (plan->ran)
      (void)hipStreamSynchronize(plan->last_stream);
    for (auto& L : plan->unpack) {
      L.d_jobs.release();
      L.d_block_start.release();
    }
    plan->d_sraw_jobs.release();
    plan->d_sraw_starts.release();
    for (auto& e : plan->events) {
      (void)hipEventDestroy(e.start);
      (void)hipEventDestroy(e.stop);
    }
    if (plan->ktimer)
      for (int i = 0; i < plan->ktimer->created; ++i)
        (void)hipEventDestroy(plan->ktimer->ev[i]);
    plan->ljpeg.reset();
    samsung_v2_plan_destroy(plan->sv2);
  }
  delete plan;
}

// ---------------------------------------------------------------------------
// Host-pointer calls: stage H2D, run a temporary plan, stage D2H.
// ---------------------------------------------------------------------------
namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Runs n unpack tiles that all write into the same host image.
int unpack_host(rsx_ctx* ctx, int n, const rsx_unpack_desc* descs,
                const uint8_t* const* ins, const size_t* in_bytes,
                const rsx_image* img, int32_t* statuses) {
  ++ctx->host_calls;
  // device-side layout: inputs back to back (16-byte aligned), one compact
  // output rectangle per tile
  std::vector<rsx_unpack_job> jobs(n);
  std::vector<int32_t> st(n, RSX_OK);
  size_t in_total = 0, out_total = 0;
  struct OutRect {
    size_t dev_off, dev_pitch, width_bytes, rows, host_off;
  };
  std::vector<OutRect> rects(n);
  for (int i = 0; i < n; ++i) {
    st[i] = validate_unpack(descs[i], *img, in_bytes[i]);
    if (st[i] != RSX_OK)
      continue;
    const rsx_unpack_desc& d = descs[i];
    const size_t strip = size_t(d.crop_h) * size_t(d.input_pitch_bytes);
    rsx_unpack_job& j = jobs[i];
    j.desc = d;
    j.in_offset = in_total;
    j.in_bytes = strip;
    in_total += align_up(strip, 16);
    const size_t cols = size_t(d.crop_w) * size_t(img->cpp);
    const int64_t rows =
        std::min<int64_t>(d.crop_h, int64_t(img->dim_y) - d.crop_y);
    const bool copy_path = d.bit_order == RSX_ORDER_LSB && d.bits_per_pixel == 16;
    OutRect r;
    r.dev_pitch = align_up(cols * 2, 16);
    r.width_bytes = cols * 2;
    r.rows = size_t(std::max<int64_t>(rows, 0));
    r.dev_off = out_total;
    r.host_off = size_t(d.crop_y) * img->pitch_bytes +
                 (copy_path ? size_t(d.crop_x) * img->cpp * 2 : 0);
    out_total += r.dev_pitch * r.rows;
    rects[i] = r;
    // The device image view is the compact rectangle: same dims as the host
    // image but re-based so that row crop_y, column 0 (or crop_x for the copy
    // path) is the rectangle's origin.
    j.img = *img;
    j.img.pitch_bytes = uint32_t(r.dev_pitch);
    j.img_offset = 0; // patched below via desc-relative offset
  }
  LaneGuard lane(ctx); // staging + stream of this call
  if (!lane.lane)
    return RSX_ERR_DEVICE;
  if (int e = lane.lane->d_in.ensure(in_total + 16))
    return e;
  if (int e = lane.lane->d_out.ensure(out_total + 16))
    return e;
  hipStream_t s = lane.lane->stream;
  auto upload_all = [&]() -> int {
    for (int i = 0; i < n; ++i) {
      if (st[i] != RSX_OK)
        continue;
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<uint8_t*>(lane.lane->d_in.ptr) +
                                            jobs[i].in_offset,
                                        ins[i], jobs[i].in_bytes,
                                        hipMemcpyHostToDevice, s));
    }
    return RSX_OK;
  };
  // Build device jobs directly (the compact rectangles are not expressible as
  // a plain image view when crop_y > 0, so bypass flatten's offset arithmetic).
  std::vector<UnpackJobDev> per_order[4];
  for (int i = 0; i < n; ++i) {
    if (st[i] != RSX_OK || rects[i].rows == 0)
      continue;
    const rsx_unpack_desc& d = descs[i];
    UnpackJobDev u{};
    u.in_offset = jobs[i].in_offset;
    u.stream_bytes = uint64_t(d.crop_h) * uint64_t(d.input_pitch_bytes);
    u.in_pitch = uint32_t(d.input_pitch_bytes);
    u.out_pitch = uint32_t(rects[i].dev_pitch);
    u.n_rows = uint32_t(rects[i].rows);
    u.cols = uint32_t(d.crop_w) * uint32_t(img->cpp);
    u.bps = uint32_t(d.bits_per_pixel);
    u.out_offset = rects[i].dev_off;
    unpack_blocks_for(&u);
    const uintptr_t a = reinterpret_cast<uintptr_t>(lane.lane->d_out.ptr) + u.out_offset;
    u.out_aligned = ((a & 15) == 0 && (u.out_pitch & 15) == 0) ? 1u : 0u;
    per_order[d.bit_order].push_back(u);
  }
  DeviceBuffer d_jobs, d_starts;
  // Large inputs (a whole frame through readUncompressedRaw, the tiles of an uncompressed
  // DNG): in row bands, the upload of band k + 1 (a helper thread, a stream of its own) under
  // the unpack and the download of band k.  PCIe is full duplex and the two directions have
  // DMA engines of their own, but a copy from or to pageable memory keeps its calling thread
  // until it is done -- so one thread per direction.  cfg 2: 78 MB up + 89 MB down took their
  // sum at 56 GB/s, 3.0 ms; banded 2.3 ms.
  {
    constexpr size_t BAND_BYTES = size_t(8) << 20, OVERLAP_MIN = size_t(16) << 20;
    constexpr int MAX_BANDS = 32;
    struct Band {
      UnpackJobDev u;
      int order;
      const uint8_t* src;
      uint8_t* dst;
      uint32_t blocks;
      size_t dev_pitch, width_bytes;
    };
    std::vector<Band> bands;
    size_t in_used_total = 0;
    for (int i = 0; i < n; ++i)
      if (st[i] == RSX_OK && rects[i].rows != 0)
        in_used_total += size_t(rects[i].rows) * size_t(descs[i].input_pitch_bytes);
    if (in_used_total >= OVERLAP_MIN && ctx->host_overlap) {
      size_t k_of_order[4] = {0, 0, 0, 0};
      for (int i = 0; i < n; ++i) {
        if (st[i] != RSX_OK || rects[i].rows == 0)
          continue;
        const rsx_unpack_desc& d = descs[i];
        const OutRect& r = rects[i];
        const UnpackJobDev base = per_order[d.bit_order][k_of_order[d.bit_order]++];
        const size_t in_used = size_t(r.rows) * size_t(d.input_pitch_bytes);
        const size_t nb = std::max<size_t>(1, std::min<size_t>(8, in_used / BAND_BYTES));
        const uint32_t rows_per = uint32_t((r.rows + nb - 1) / nb);
        for (uint32_t r0 = 0; r0 < r.rows; r0 += rows_per) {
          Band bd;
          bd.u = base;
          bd.u.n_rows = uint32_t(std::min<size_t>(rows_per, r.rows - r0));
          bd.u.in_offset = base.in_offset + uint64_t(r0) * base.in_pitch;
          bd.u.out_offset = base.out_offset + uint64_t(r0) * base.out_pitch;
          bd.u.stream_bytes = uint64_t(bd.u.n_rows) * base.in_pitch;
          bd.blocks = unpack_blocks_for(&bd.u);
          bd.order = d.bit_order;
          bd.src = ins[i] + size_t(r0) * base.in_pitch;
          bd.dst = static_cast<uint8_t*>(img->data) + r.host_off + size_t(r0) * img->pitch_bytes;
          bd.dev_pitch = r.dev_pitch;
          bd.width_bytes = r.width_bytes;
          bands.push_back(bd);
        }
      }
    }
    const int nb = int(bands.size());
    if (nb >= 2 && nb <= MAX_BANDS && lane.lane->ensure_overlap(nb)) {
      std::vector<UnpackJobDev> bj(nb);
      std::vector<uint32_t> starts(2 * size_t(nb));
      for (int b = 0; b < nb; ++b) {
        bj[b] = bands[b].u;
        starts[2 * b] = 0;
        starts[2 * b + 1] = bands[b].blocks;
      }
      if (int e = d_jobs.ensure(bj.size() * sizeof(UnpackJobDev)))
        return e;
      if (int e = d_starts.ensure(starts.size() * sizeof(uint32_t)))
        return e;
      {
        hipError_t e = hipMemcpyAsync(d_jobs.ptr, bj.data(), bj.size() * sizeof(UnpackJobDev),
                                      hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
          e = hipMemcpyAsync(d_starts.ptr, starts.data(), starts.size() * sizeof(uint32_t),
                             hipMemcpyHostToDevice, s);
        if (e != hipSuccess) {
          (void)hipStreamSynchronize(s); // (bj / starts / the device temporaries outlive the copies)
          set_error(ctx, std::string("unpack (banded): ") + hipGetErrorString(e));
          return RSX_ERR_DEVICE;
        }
      }
      // progress of the uploader: bands whose copy + event are queued (under a mutex, the
      // caller sleeps on the condition variable)
      struct Progress {
        std::mutex m;
        std::condition_variable cv;
        int ready = 0;
        bool failed = false;
      } prog;
      uint8_t* d_in_base = static_cast<uint8_t*>(lane.lane->d_in.ptr);
      hipStream_t s_up = lane.lane->stream_up;
      const int device = ctx->device;
      auto upload_bands = [&]() {
        bool ok = hipSetDevice(device) == hipSuccess;
        for (int b = 0; b < nb; ++b) {
          if (ok) {
            hipError_t e = hipMemcpyAsync(d_in_base + bands[b].u.in_offset, bands[b].src,
                                          size_t(bands[b].u.stream_bytes), hipMemcpyHostToDevice,
                                          s_up);
            if (e == hipSuccess)
              e = hipEventRecord(lane.lane->ev_up[b], s_up);
            ok = e == hipSuccess;
          }
          {
            std::lock_guard<std::mutex> g(prog.m);
            prog.ready = b + 1;
            prog.failed = prog.failed || !ok;
          }
          prog.cv.notify_all();
        }
      };
      // (a persistent helper of the context; none to be had: the uploads first, then the rest)
      rsx::HelperPool::Handle uploader = ctx->helpers.submit(upload_bands);
      if (!uploader)
        upload_bands();
      hipError_t err = hipSuccess;
      bool up_failed = false;
      for (int b = 0; b < nb && err == hipSuccess; ++b) {
        {
          std::unique_lock<std::mutex> g(prog.m);
          prog.cv.wait(g, [&]() { return prog.ready > b; });
          up_failed = prog.failed;
        }
        if (up_failed)
          break;
        err = hipStreamWaitEvent(s, lane.lane->ev_up[b], 0);
        if (err == hipSuccess)
          err = launch_unpack(bands[b].order, static_cast<UnpackJobDev*>(d_jobs.ptr) + b,
                              static_cast<uint32_t*>(d_starts.ptr) + 2 * b, 1, bands[b].blocks,
                              lane.lane->d_in.ptr, lane.lane->d_out.ptr, s);
        if (err == hipSuccess) {
          // (download_rects: on the grid straight into the image, the rest through the lane's
          // page-locked staging; it returns with the band in the caller's memory -- the
          // uploader thread keeps the other direction of the link busy meanwhile)
          const DownRect dr{bands[b].dst, img->pitch_bytes,
                            static_cast<uint8_t*>(lane.lane->d_out.ptr) + bands[b].u.out_offset,
                            bands[b].dev_pitch, bands[b].width_bytes, bands[b].u.n_rows};
          const bool grid = on_grid(reinterpret_cast<uintptr_t>(dr.host)) && on_grid(dr.host_pitch) &&
                            on_grid(dr.bytes);
          if (grid) // (the usual frame: queued, not waited for -- band b + 1 is launched under it)
            err = hipMemcpy2DAsync(dr.host, dr.host_pitch, dr.dev, dr.dev_pitch, dr.bytes, dr.rows,
                                   hipMemcpyDeviceToHost, s);
          else if (download_rects(ctx, lane.lane, s, &dr, 1) != RSX_OK)
            err = hipErrorUnknown;
        }
      }
      ctx->helpers.wait(uploader);
      {
        std::lock_guard<std::mutex> g(prog.m);
        up_failed = prog.failed;
      }
      // BOTH streams drained on every way out, failed or not: the caller's input (s_up reads
      // it) and image (s writes it) may be freed the moment this call returns, and the
      // temporaries below are released
      const hipError_t e_up = hipStreamSynchronize(s_up), e_s = hipStreamSynchronize(s);
      if (err == hipSuccess)
        err = e_up != hipSuccess ? e_up : e_s;
      d_jobs.release();
      d_starts.release();
      if (err != hipSuccess || up_failed) {
        set_error(ctx, std::string("unpack (banded): ") +
                           (err != hipSuccess ? hipGetErrorString(err) : "upload failed"));
        return RSX_ERR_DEVICE;
      }
      int rc = RSX_OK;
      for (int i = 0; i < n; ++i) {
        if (statuses)
          statuses[i] = st[i];
        if (st[i] != RSX_OK)
          rc = st[i];
      }
      return rc;
    }
  }
  if (int e = upload_all())
    return e;
  for (int order = 0; order < 4; ++order) {
    auto& v = per_order[order];
    if (v.empty())
      continue;
    std::vector<uint32_t> starts(v.size() + 1, 0);
    for (size_t k = 0; k < v.size(); ++k)
      starts[k + 1] = starts[k] + v[k].n_rows * v[k].segs_per_row;
    // stream-ordered: sync before the temporaries are reused by the next order
    if (int e = d_jobs.ensure(v.size() * sizeof(UnpackJobDev)))
      return e;
    if (int e = d_starts.ensure(starts.size() * sizeof(uint32_t)))
      return e;
    RSX_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs.ptr, v.data(),
                                      v.size() * sizeof(UnpackJobDev),
                                      hipMemcpyHostToDevice, s));
    RSX_HIP_CHECK(ctx, hipMemcpyAsync(d_starts.ptr, starts.data(),
                                      starts.size() * sizeof(uint32_t),
                                      hipMemcpyHostToDevice, s));
    RSX_HIP_CHECK(ctx, launch_unpack(order, static_cast<UnpackJobDev*>(d_jobs.ptr),
                                     static_cast<uint32_t*>(d_starts.ptr),
                                     int(v.size()), starts.back(), lane.lane->d_in.ptr,
                                     lane.lane->d_out.ptr, s));
    RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
  }
  {
    std::vector<DownRect> down;
    for (int i = 0; i < n; ++i) {
      if (st[i] != RSX_OK || rects[i].rows == 0)
        continue;
      const OutRect& r = rects[i];
      down.push_back({static_cast<uint8_t*>(img->data) + r.host_off, img->pitch_bytes,
                      static_cast<uint8_t*>(lane.lane->d_out.ptr) + r.dev_off, r.dev_pitch,
                      r.width_bytes, r.rows});
    }
    if (int e = download_rects(ctx, lane.lane, s, down.data(), down.size()))
      return e;
  }
  d_jobs.release();
  d_starts.release();
  int rc = RSX_OK;
  for (int i = 0; i < n; ++i) {
    if (statuses)
      statuses[i] = st[i];
    if (st[i] != RSX_OK)
      rc = st[i];
  }
  return rc;
}

} // namespace

extern "C" int rsx_unpack_u16(rsx_ctx* ctx, const rsx_unpack_desc* d,
                              const uint8_t* in, size_t in_bytes,
                              const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return unpack_host(ctx, 1, d, &in, &in_bytes, img, nullptr);
}

extern "C" int rsx_unpack_f32(rsx_ctx* ctx, const rsx_unpack_desc* d, const uint8_t* in,
                              size_t in_bytes, const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int st = validate_unpack_f32(*d, *img, in_bytes))
    return st;
  const size_t used = size_t(d->crop_h) * size_t(d->input_pitch_bytes);
  const int64_t rows = std::min<int64_t>(d->crop_h, int64_t(img->dim_y) - d->crop_y);
  if (rows <= 0)
    return RSX_OK;
  // device image = the compact rectangle that is written
  const size_t width_bytes = size_t(d->crop_w) * img->cpp * 4;
  rsx_unpack_job job{};
  job.desc = *d;
  job.desc.crop_x = 0;
  job.desc.crop_y = 0;
  job.in_bytes = used;
  job.img = *img;
  job.img.dim_y = int32_t(rows);
  job.img.pitch_bytes = uint32_t(align_up(width_bytes, 16));
  LaneGuard lane(ctx); // staging + stream of this call
  if (!lane.lane)
    return RSX_ERR_DEVICE;
  if (int e = lane.lane->d_in.ensure(used + 16))
    return e;
  if (int e = lane.lane->d_out.ensure(size_t(job.img.pitch_bytes) * size_t(rows) + 16))
    return e;
  hipStream_t s = lane.lane->stream;
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(lane.lane->d_in.ptr, in, used, hipMemcpyHostToDevice, s));
  rsx_plan* plan = nullptr;
  if (int st = rsx_unpack_f32_plan_create(ctx, 1, &job, &plan))
    return st;
  int rc = rsx_plan_run(plan, lane.lane->d_in.ptr, lane.lane->d_out.ptr, s);
  if (rc == RSX_OK) {
    const size_t x0 = d->bits_per_pixel == 32 ? size_t(d->crop_x) * img->cpp
                                              : size_t(d->crop_x);
    uint8_t* dst = static_cast<uint8_t*>(img->data) +
                   size_t(d->crop_y) * img->pitch_bytes + x0 * 4;
    const DownRect dr{dst, img->pitch_bytes, static_cast<uint8_t*>(lane.lane->d_out.ptr),
                      job.img.pitch_bytes, width_bytes, size_t(rows)};
    rc = download_rects(ctx, lane.lane, s, &dr, 1);
  }
  rsx_plan_destroy(plan);
  return rc;
}

extern "C" int rsx_unpack_variant_u16(rsx_ctx* ctx, const rsx_unpack_variant_desc* d,
                                      const uint8_t* in, size_t in_bytes,
                                      const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int st = validate_unpack_variant(*d, *img, in_bytes))
    return st;
  uint64_t bpl = 0;
  unpack_variant_bytes_per_line(*d, &bpl);
  const size_t used = size_t(bpl) * size_t(d->h);
  // device image = the compact w x h rectangle at the image origin
  rsx_unpack_variant_job job{};
  job.desc = *d;
  job.in_offset = 0;
  job.in_bytes = used;
  job.img_offset = 0;
  job.img = *img;
  job.img.pitch_bytes = uint32_t(align_up(size_t(d->w) * 2, 16));
  const size_t out_bytes = size_t(job.img.pitch_bytes) * size_t(d->h);
  LaneGuard lane(ctx); // staging + stream of this call
  if (!lane.lane)
    return RSX_ERR_DEVICE;
  if (int e = lane.lane->d_in.ensure(used + 16))
    return e;
  if (int e = lane.lane->d_out.ensure(out_bytes + 16))
    return e;
  hipStream_t s = lane.lane->stream;
  RSX_HIP_CHECK(ctx, hipMemcpyAsync(lane.lane->d_in.ptr, in, used, hipMemcpyHostToDevice, s));
  rsx_plan* plan = nullptr;
  if (int st = rsx_unpack_variant_plan_create(ctx, 1, &job, &plan))
    return st;
  int rc = rsx_plan_run(plan, lane.lane->d_in.ptr, lane.lane->d_out.ptr, s);
  if (rc == RSX_OK) {
    const DownRect dr{static_cast<uint8_t*>(img->data), img->pitch_bytes,
                      static_cast<uint8_t*>(lane.lane->d_out.ptr), job.img.pitch_bytes,
                      size_t(d->w) * 2, size_t(d->h)};
    rc = download_rects(ctx, lane.lane, s, &dr, 1);
  }
  rsx_plan_destroy(plan);
  return rc;
}

extern "C" int rsx_dng_decompress_uncompressed(rsx_ctx* ctx, int n_tiles,
                                               const rsx_dng_unpack_tile* tiles,
                                               const rsx_image* img,
                                               int32_t* tile_status) {
  if (!ctx || !tiles || n_tiles < 1 || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<rsx_unpack_desc> descs(n_tiles);
  std::vector<const uint8_t*> ins(n_tiles);
  std::vector<size_t> sizes(n_tiles);
  for (int i = 0; i < n_tiles; ++i) {
    descs[i] = tiles[i].desc;
    ins[i] = tiles[i].in;
    sizes[i] = tiles[i].in_bytes;
  }
  std::vector<int32_t> st(n_tiles, RSX_OK);
  const int rc = unpack_host(ctx, n_tiles, descs.data(), ins.data(), sizes.data(),
                             img, st.data());
  if (tile_status)
    std::copy(st.begin(), st.end(), tile_status);
  if (rc == RSX_ERR_DEVICE || rc == RSX_ERR_NOMEM)
    return rc;
  for (int i = 0; i < n_tiles; ++i)
    if (st[i] != RSX_OK)
      return RSX_ERR_TILE_ERRORS; // AbstractDngDecompressor.cpp:247-251
  return RSX_OK;
}

// ---------------------------------------------------------------------------
// LJPEG / CR2 entry points: see rsx_ljpeg.hip for the implementation.
// ---------------------------------------------------------------------------
namespace {
// Tables with the same CONTENTS are one table.  The reference binds a decoder per DHT
// slot (AbstractLJpegDecoder.h:112-125) and the shim de-duplicates by object address; a
// file that declares the same code twice (Canon's Th = 0 / Th = 1, many DNG writers) must
// not look like a two-table stream to the kernels: one table keeps the component phase
// out of the synchronisation state and takes the single-pass kernel.
struct UniqueTables {
  rsx_huff_table t[RSX_MAX_COMPONENTS];
};
bool same_table(const rsx_huff_table& a, const rsx_huff_table& b) {
  return a.n_code_values == b.n_code_values && a.fix_dng_bug16 == b.fix_dng_bug16 &&
         std::memcmp(a.n_codes_per_length, b.n_codes_per_length, 16) == 0 &&
         std::memcmp(a.code_values, b.code_values, a.n_code_values) == 0;
}
void dedupe_tables(LJpegJobIn& in, UniqueTables& store) {
  if (in.status != RSX_OK || in.n_tables <= 1 || in.n_tables > RSX_MAX_COMPONENTS)
    return;
  int map[RSX_MAX_COMPONENTS], n = 0;
  for (int t = 0; t < in.n_tables; ++t) {
    int u = 0;
    while (u < n && !same_table(store.t[u], in.tables[t]))
      ++u;
    if (u == n)
      store.t[n++] = in.tables[t];
    map[t] = u;
  }
  if (n == in.n_tables)
    return;
  for (uint8_t& c : in.geom.comp_of_phase)
    if (int(c) < in.n_tables)
      c = uint8_t(map[c]);
  in.tables = store.t;
  in.n_tables = n;
}
} // namespace

extern "C" int rsx_ljpeg_plan_create(rsx_ctx* ctx, int n_jobs,
                                     const rsx_ljpeg_job* jobs,
                                     rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  std::vector<UniqueTables> unique(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    in[i].status = build_ljpeg_stream(jobs[i].desc, jobs[i].img, &in[i].geom);
    in[i].geom.in_offset = jobs[i].in_offset;
    in[i].geom.in_bytes = jobs[i].in_bytes;
    in[i].geom.img_offset = jobs[i].img_offset;
    in[i].tables = jobs[i].desc.tables;
    in[i].n_tables = jobs[i].desc.n_tables;
    in[i].rows_per_restart_interval = jobs[i].desc.rows_per_restart_interval;
    in[i].frame_h = jobs[i].desc.frame_h;
    dedupe_tables(in[i], unique[i]);
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_cr2_plan_create(rsx_ctx* ctx, int n_jobs,
                                   const rsx_cr2_job* jobs, rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  std::vector<UniqueTables> unique(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    in[i].status = build_cr2_stream(jobs[i].desc, jobs[i].img, &in[i].geom);
    in[i].geom.in_offset = jobs[i].in_offset;
    in[i].geom.in_bytes = jobs[i].in_bytes;
    in[i].geom.img_offset = jobs[i].img_offset;
    in[i].tables = jobs[i].desc.tables;
    in[i].n_tables = jobs[i].desc.n_tables;
    in[i].rows_per_restart_interval = 0; // CR2 rejects DRI (Cr2LJpegDecoder.cpp:59-60)
    in[i].frame_h = 0;
    dedupe_tables(in[i], unique[i]);
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

// ---------------------------------------------------------------------------
// SamsungV2Decompressor
// ---------------------------------------------------------------------------
extern "C" int rsx_samsung_v2_validate(const rsx_samsung_v2_desc* d, const rsx_image* img) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return samsung_v2_validate(*d, *img);
}

extern "C" int rsx_samsung_v2_plan_create(rsx_ctx* ctx, int n_jobs,
                                          const rsx_samsung_v2_job* jobs,
                                          rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_SV2;
  plan->n_jobs = n_jobs;
  plan->job_status.assign(n_jobs, RSX_OK);
  if (int st = samsung_v2_plan_create(ctx, n_jobs, jobs, &plan->sv2))
    return st;
  *out_plan = plan.release();
  return RSX_OK;
}

// ---------------------------------------------------------------------------
// Cr2sRawInterpolator
// ---------------------------------------------------------------------------
extern "C" int rsx_sraw_validate(const rsx_sraw_desc* d, const rsx_image* in,
                                 const rsx_image* out) {
  if (!d || !in || !out)
    return RSX_ERR_INVALID_ARG;
  return validate_sraw(*d, *in, *out);
}

extern "C" int rsx_sraw_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_sraw_job* jobs,
                                    rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_SRAW;
  plan->n_jobs = n_jobs;
  plan->job_status.assign(n_jobs, RSX_OK);
  std::vector<SrawJobDev> v;
  std::vector<uint32_t> starts(1, 0);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_sraw_job& j = jobs[i];
    int st = validate_sraw(j.desc, j.in, j.img);
    // the kernel moves 16-byte pieces: RawImage rows are (pitch = roundUp(.., 16),
    // common/RawImage.cpp:80-83), the image bases must be as well
    if (st == RSX_OK && (j.in.pitch_bytes % 16 != 0 || j.img.pitch_bytes % 16 != 0 ||
                         j.in_offset % 16 != 0 || j.img_offset % 16 != 0))
      st = RSX_ERR_INVALID_ARG;
    plan->job_status[i] = st;
    if (st != RSX_OK)
      continue;
    SrawJobDev d{};
    d.in_offset = j.in_offset;
    d.out_offset = j.img_offset;
    d.in_pitch = j.in.pitch_bytes;
    d.out_pitch = j.img.pitch_bytes;
    d.rows = uint32_t(j.in.dim_y);
    d.gs = uint32_t(2 + 2 * j.desc.subsampling_y);
    d.num_mcus = uint32_t(j.in.dim_x) / d.gs;
    d.version = uint32_t(j.desc.version);
    for (int k = 0; k < 3; ++k)
      d.coeffs[k] = j.desc.sraw_coeffs[k];
    d.hue = j.desc.hue;
    starts.push_back(starts.back() + sraw_blocks_for(&d));
    v.push_back(d);
    plan->sraw_versions[j.desc.version] = true;
  }
  plan->n_sraw = int(v.size());
  plan->sraw_blocks = starts.back();
  if (int st = upload(ctx, plan->d_sraw_jobs, v.data(), v.size() * sizeof(SrawJobDev)))
    return st;
  if (int st = upload(ctx, plan->d_sraw_starts, starts.data(), starts.size() * 4))
    return st;
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_sraw_interpolate(rsx_ctx* ctx, const rsx_sraw_desc* d,
                                    const rsx_image* in, const rsx_image* out) {
  if (!ctx || !d || !in || !out || !in->data || !out->data)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int st = validate_sraw(*d, *in, *out))
    return st;
  // compact device images (row padding never travels)
  rsx_sraw_job job{};
  job.desc = *d;
  job.in = *in;
  job.img = *out;
  const size_t in_w = size_t(in->dim_x) * 2, out_w = size_t(out->dim_x) * 3 * 2;
  job.in.pitch_bytes = uint32_t(align_up(in_w, 16));
  job.img.pitch_bytes = uint32_t(align_up(out_w, 16));
  const size_t in_bytes = size_t(job.in.pitch_bytes) * in->dim_y;
  const size_t out_bytes = size_t(job.img.pitch_bytes) * out->dim_y;
  LaneGuard lane(ctx); // staging + stream of this call
  if (!lane.lane)
    return RSX_ERR_DEVICE;
  if (int e = lane.lane->d_in.ensure(in_bytes + 16))
    return e;
  if (int e = lane.lane->d_out.ensure(out_bytes + 16))
    return e;
  hipStream_t s = lane.lane->stream;
  RSX_HIP_CHECK(ctx, hipMemcpy2DAsync(lane.lane->d_in.ptr, job.in.pitch_bytes, in->data,
                                      in->pitch_bytes, in_w, size_t(in->dim_y),
                                      hipMemcpyHostToDevice, s));
  rsx_plan* plan = nullptr;
  if (int st = rsx_sraw_plan_create(ctx, 1, &job, &plan))
    return st;
  int rc = rsx_plan_run(plan, lane.lane->d_in.ptr, lane.lane->d_out.ptr, s);
  if (rc == RSX_OK) {
    const DownRect dr{static_cast<uint8_t*>(out->data), out->pitch_bytes,
                      static_cast<uint8_t*>(lane.lane->d_out.ptr), job.img.pitch_bytes, out_w,
                      size_t(out->dim_y)};
    rc = download_rects(ctx, lane.lane, s, &dr, 1);
  }
  rsx_plan_destroy(plan);
  return rc;
}

extern "C" int rsx_nikon_validate(const rsx_nikon_desc* d, const rsx_image* img) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_nikon(*d, *img);
}

extern "C" int rsx_nikon_plan_create(rsx_ctx* ctx, int n_jobs,
                                     const rsx_nikon_job* jobs, rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_nikon_desc& d = jobs[i].desc;
    const rsx_image& img = jobs[i].img;
    LJpegJobIn& J = in[i];
    J.status = validate_nikon(d, img);
    if (J.status == RSX_OK && img.pitch_bytes % 2 != 0)
      J.status = RSX_ERR_INVALID_ARG;
    if (J.status != RSX_OK)
      continue;
    StreamGeom& g = J.geom;
    std::memset(&g, 0, sizeof g);
    g.kind = 2;
    g.raw = 1;
    g.in_offset = jobs[i].in_offset;
    g.in_bytes = jobs[i].in_bytes;
    g.img_offset = jobs[i].img_offset;
    g.img_pitch_bytes = img.pitch_bytes;
    // one table for the whole stream: the two columns alternate (:525-527) but
    // share it, so the "component" only matters to the reconstruction
    g.n_comp = 2;
    g.period = 2;
    g.rows = uint32_t(d.split ? d.split : img.dim_y); // decompress(bits, 0, split) :555-556
    g.row_samples = uint32_t(img.dim_x);
    g.mcu_w = g.mcu_h = 1;
    g.keep_samples = g.row_samples;
    J.tables = &d.tables[0];
    J.n_tables = 1;
    NikonIn& N = J.nikon;
    for (int k = 0; k < 4; ++k)
      N.p_up[k] = (&d.p_up[0][0])[k];
    N.uncorrected = d.uncorrected_raw_values != 0;
    N.split = d.split;
    N.height = img.dim_y;
    N.seed_offset = jobs[i].in_offset;
    if (d.split)
      N.table_after_split = d.tables[1];
    if (!N.uncorrected)
      build_dither_table(d.curve, d.curve_size, &N.dither);
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_pentax_validate(const rsx_pentax_desc* d, const rsx_image* img) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_pentax(*d, *img);
}

extern "C" int rsx_pentax_plan_create(rsx_ctx* ctx, int n_jobs,
                                      const rsx_pentax_job* jobs, rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_image& img = jobs[i].img;
    LJpegJobIn& J = in[i];
    J.status = validate_pentax(jobs[i].desc, img);
    if (J.status == RSX_OK && img.pitch_bytes % 2 != 0)
      J.status = RSX_ERR_INVALID_ARG;
    if (J.status != RSX_OK)
      continue;
    StreamGeom& g = J.geom;
    std::memset(&g, 0, sizeof g);
    g.kind = 2; // the Nikon reconstruction kernels, Pentax flavour
    g.raw = 1;
    g.in_offset = jobs[i].in_offset;
    g.in_bytes = jobs[i].in_bytes;
    g.img_offset = jobs[i].img_offset;
    g.img_pitch_bytes = img.pitch_bytes;
    g.n_comp = 2;
    g.period = 2;
    g.rows = uint32_t(img.dim_y);
    g.row_samples = uint32_t(img.dim_x);
    g.mcu_w = g.mcu_h = 1;
    g.keep_samples = g.row_samples;
    J.tables = &jobs[i].desc.table;
    J.n_tables = 1;
    J.nikon.uncorrected = true;
    J.nikon.pentax = true; // predictors start at 0, range check instead of clamp
    J.nikon.height = img.dim_y;
    J.nikon.seed_offset = jobs[i].in_offset;
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_hasselblad_validate(const rsx_hasselblad_desc* d, const rsx_image* img) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_hasselblad(*d, *img);
}

extern "C" int rsx_hasselblad_plan_create(rsx_ctx* ctx, int n_jobs,
                                          const rsx_hasselblad_job* jobs,
                                          rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_image& img = jobs[i].img;
    LJpegJobIn& J = in[i];
    J.status = validate_hasselblad(jobs[i].desc, img);
    if (J.status == RSX_OK && img.pitch_bytes % 2 != 0)
      J.status = RSX_ERR_INVALID_ARG;
    if (J.status != RSX_OK)
      continue;
    StreamGeom& g = J.geom;
    std::memset(&g, 0, sizeof g);
    g.kind = 0; // rows of W samples written in place, like an LJPEG tile that is the image
    g.raw = 1;  // BitStreamerMSB32: no stuffing, no markers, 8-byte over-read budget
    g.pair = 1;
    g.no_vertical = 1; // "int p1 = rec.initPred; int p2 = rec.initPred;" per row (:83-85)
    g.in_offset = jobs[i].in_offset;
    g.in_bytes = jobs[i].in_bytes;
    g.img_offset = jobs[i].img_offset;
    g.img_pitch_bytes = img.pitch_bytes;
    g.n_comp = 2; // p1 / p2 alternate along the row (:86-96)
    g.period = 2;
    g.pred_of_phase[0] = 0;
    g.pred_of_phase[1] = 1;
    g.init_pred[0] = g.init_pred[1] = jobs[i].desc.init_pred;
    g.seed_pos[0] = 0;
    g.seed_pos[1] = 1;
    g.rows = uint32_t(img.dim_y);
    g.row_samples = uint32_t(img.dim_x);
    g.mcu_w = 2;
    g.mcu_h = 1;
    g.keep_samples = g.row_samples;
    J.tables = &jobs[i].desc.table;
    J.n_tables = 1;
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_samsung_v1_validate(const rsx_samsung_v1_desc* d, const rsx_image* img) {
  if (!d || !img)
    return RSX_ERR_INVALID_ARG;
  return validate_samsung_v1(*d, *img);
}

extern "C" int rsx_samsung_v1_plan_create(rsx_ctx* ctx, int n_jobs,
                                          const rsx_samsung_v1_job* jobs,
                                          rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_image& img = jobs[i].img;
    LJpegJobIn& J = in[i];
    J.status = validate_samsung_v1(jobs[i].desc, img);
    if (J.status == RSX_OK && img.pitch_bytes % 2 != 0)
      J.status = RSX_ERR_INVALID_ARG;
    if (J.status != RSX_OK)
      continue;
    StreamGeom& g = J.geom;
    std::memset(&g, 0, sizeof g);
    g.kind = 2; // the Nikon / Pentax reconstruction kernels
    g.raw = 1;
    g.in_offset = jobs[i].in_offset;
    g.in_bytes = jobs[i].in_bytes;
    g.img_offset = jobs[i].img_offset;
    g.img_pitch_bytes = img.pitch_bytes;
    g.n_comp = 2;
    g.period = 2;
    g.rows = uint32_t(img.dim_y);
    g.row_samples = uint32_t(img.dim_x);
    g.mcu_w = g.mcu_h = 1;
    g.keep_samples = g.row_samples;
    // samsungDiff refills with fill(23), not fill(32) (.cpp:66): a symbol at bit c
    // needs 32 K - c >= 23, so symbols may start 9 bits later than with fill(32)
    g.raw_limit = 32 * ((uint64_t(jobs[i].in_bytes) + 8) / 4) + 9 + 1;
    J.n_tables = 1;
    J.explicit_enc_len = jobs[i].desc.enc_len;
    J.explicit_diff_len = jobs[i].desc.diff_len;
    J.explicit_n = jobs[i].desc.n_entries;
    J.nikon.uncorrected = true;
    J.nikon.pentax = true;
    J.nikon.range_bits = jobs[i].desc.bits;
    J.nikon.height = img.dim_y;
    J.nikon.seed_offset = jobs[i].in_offset;
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

extern "C" int rsx_sony_arw1_validate(const rsx_image* img) {
  if (!img)
    return RSX_ERR_INVALID_ARG;
  return validate_sony_arw1(*img);
}

namespace {
// SonyArw1Decompressor.cpp:76-83 as (code length, difference length) pairs in
// ascending code order of an 11-bit table: "00" + k zeros + "1" = 4 + k for
// k = 8 .. 0, "010" = 3, "011" = 0, "10" = 2, "11" = 1.  The first slot is the
// prefix of the lengths 13..17 (codes of 12..15 bits): such a difference is at
// least 4096 in magnitude, so the value always leaves 0..4095 and the
// reference throws (.cpp:88-89) -- an invalid code here, reported the same way.
constexpr uint8_t SONY_ENC[14] = {11, 11, 10, 9, 8, 7, 6, 5, 4, 3, 3, 3, 2, 2};
constexpr uint8_t SONY_DIF[14] = {0xFF, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 0, 2, 1};
} // namespace

extern "C" int rsx_sony_arw1_plan_create(rsx_ctx* ctx, int n_jobs,
                                         const rsx_sony_arw1_job* jobs,
                                         rsx_plan** out_plan) {
  if (!ctx || !jobs || n_jobs < 1 || !out_plan)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto plan = std::make_unique<rsx_plan>();
  plan->ctx = ctx;
  plan->kind = PLAN_LJPEG;
  plan->n_jobs = n_jobs;
  std::vector<LJpegJobIn> in(n_jobs);
  for (int i = 0; i < n_jobs; ++i) {
    const rsx_image& img = jobs[i].img;
    LJpegJobIn& J = in[i];
    J.status = validate_sony_arw1(img);
    if (J.status == RSX_OK && img.pitch_bytes % 2 != 0)
      J.status = RSX_ERR_INVALID_ARG;
    if (J.status != RSX_OK)
      continue;
    StreamGeom& g = J.geom;
    std::memset(&g, 0, sizeof g);
    g.kind = 2; // int32 sums + range check, like Pentax; sony_* reconstruction kernels
    g.raw = 1;  // BitStreamerMSB, fill(32) per pixel (.cpp:70)
    g.in_offset = jobs[i].in_offset;
    g.in_bytes = jobs[i].in_bytes;
    g.img_offset = jobs[i].img_offset;
    g.img_pitch_bytes = img.pitch_bytes;
    g.n_comp = 2;
    g.period = 2;
    // a stream row is an image column (.cpp:68-69): W rows of H samples
    g.rows = uint32_t(img.dim_x);
    g.row_samples = uint32_t(img.dim_y);
    g.mcu_w = g.mcu_h = 1;
    g.keep_samples = g.row_samples;
    J.n_tables = 1;
    J.explicit_enc_len = SONY_ENC;
    J.explicit_diff_len = SONY_DIF;
    J.explicit_n = 14;
    J.explicit_bits = 11;
    J.nikon.uncorrected = true;
    J.nikon.pentax = true;
    J.nikon.range_bits = 12;
    J.nikon.sony = true;
    J.nikon.height = img.dim_x;
    J.nikon.seed_offset = jobs[i].in_offset;
  }
  LJpegPlan* lp = nullptr;
  if (int st = ljpeg_plan_create(ctx, in, &lp))
    return st;
  plan->ljpeg.reset(lp);
  *out_plan = plan.release();
  return RSX_OK;
}

namespace {

// Generic host-pointer runner for LJPEG-family jobs sharing one host image.
// rectangle of the host image a successful job has written (bytes)
struct HostRect {
  size_t row0, rows, byte0, bytes;
};
HostRect out_rect(const rsx_ljpeg_job& j) {
  return {size_t(j.desc.tile_y), size_t(j.desc.tile_h),
          size_t(j.desc.tile_x) * j.img.cpp * 2, size_t(j.desc.tile_w) * j.img.cpp * 2};
}
HostRect out_rect(const rsx_cr2_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_nikon_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_pentax_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_samsung_v1_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_samsung_v2_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_sony_arw1_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}
HostRect out_rect(const rsx_hasselblad_job& j) {
  return {0, size_t(j.img.dim_y), 0, size_t(j.img.dim_x) * 2};
}

// What a job's plan is made FROM beyond the bytes of its struct: data behind host pointers.
// The plan-cache key holds the contents, never the address (a freed and re-allocated
// curve at the same address, or one mutated in place, must not find the old plan).
template <typename JobT>
void key_job(std::vector<uint8_t>& key, const JobT& job) {
  JobT j = job;
  j.img.data = nullptr;
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&j);
  key.insert(key.end(), p, p + sizeof(JobT));
}
template <>
void key_job<rsx_nikon_job>(std::vector<uint8_t>& key, const rsx_nikon_job& job) {
  rsx_nikon_job j = job;
  j.img.data = nullptr;
  j.desc.curve = nullptr;
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&j);
  key.insert(key.end(), p, p + sizeof j);
  if (job.desc.curve && job.desc.curve_size > 0 && job.desc.curve_size <= 65536) {
    const uint8_t* c = reinterpret_cast<const uint8_t*>(job.desc.curve);
    key.insert(key.end(), c, c + size_t(job.desc.curve_size) * sizeof(uint16_t));
  }
}

// One large entropy-coded stream through a host-pointer call, in CHUNKS (round 6; the review's
// "download under decode").  Until now: the whole input up, the kernels, the whole image down,
// one after the other -- 0.55 + 0.1 + 1.07 ms for a 6720 x 4480 CR2 frame, on a link that is full
// duplex.  K0 and the single-pass kernel need nothing of a stream but the bytes of the
// workgroups they run on (and what the workgroups in front of them left), so the plan's blocks
// go in four launches, each behind the upload of ITS quarter of the bytes (a helper thread of the
// context, a stream of its own: a copy from pageable memory keeps its calling thread), and
// what a launch completes -- whole stream rows, whole rows of a CR2 strip -- comes down while
// the next quarter is on its way up and decodes.  The rectangles go through download_rects
// like every other download.  If the stream leaves the single-pass kernel after all (a second
// pass rewrites pixels), everything comes down once more at the end.
// CHUNKED_NOT_TAKEN: nothing done, the caller takes the plain way.
constexpr int CHUNKED_NOT_TAKEN = -1000;
int ljpeg_chunked_host(rsx_ctx* ctx, rsx_ctx::HostLane* L, rsx_plan* plan, size_t in_bytes,
                       size_t in_total, const uint8_t* in, const rsx_image* img,
                       const HostRect& whole, size_t out_skip, int32_t* status,
                       uint32_t* consumed) {
  constexpr int NCH = 4;
  LJpegPlan* lp = plan->ljpeg.get();
  const uint32_t nblk = ljpeg_plan_blocks(lp);
  if (nblk < 4 * NCH || !L->ensure_overlap(NCH))
    return CHUNKED_NOT_TAKEN;
  ++ctx->chunked_calls;
  hipStream_t s = L->stream, s_up = L->stream_up;
  uint8_t* d_in = static_cast<uint8_t*>(L->d_in.ptr);
  uint8_t* const out_row0 = static_cast<uint8_t*>(L->d_out.ptr) - out_skip;
  uint32_t blk_end[NCH];
  size_t byte_end[NCH];
  for (int c = 0; c < NCH; ++c) {
    blk_end[c] = uint32_t(uint64_t(nblk) * uint32_t(c + 1) / NCH);
    // (a workgroup reads its 255 slots, the slot in front of them and 16 bytes behind)
    byte_end[c] = c + 1 == NCH ? in_bytes : std::min(in_bytes, size_t(blk_end[c]) * LJ_R + 256);
  }
  struct Progress {
    std::mutex m;
    std::condition_variable cv;
    int ready = 0;
    bool failed = false;
  } prog;
  const int device = ctx->device;
  auto upload = [&]() {
    bool ok = hipSetDevice(device) == hipSuccess;
    std::lock_guard<std::mutex> up(ctx->upload_mu);
    // (the slack behind the input: zeros, as the plain way leaves them)
    if (ok)
      ok = hipMemsetAsync(d_in + (in_bytes & ~size_t(15)), 0, in_total + 64 - (in_bytes & ~size_t(15)), s_up) ==
           hipSuccess;
    size_t from = 0;
    for (int c = 0; c < NCH; ++c) {
      if (ok && byte_end[c] > from)
        ok = hipMemcpyAsync(d_in + from, in + from, byte_end[c] - from, hipMemcpyHostToDevice, s_up) ==
             hipSuccess;
      if (ok)
        ok = hipEventRecord(L->ev_up[c], s_up) == hipSuccess;
      from = std::max(from, byte_end[c]);
      {
        std::lock_guard<std::mutex> g(prog.m);
        prog.ready = c + 1;
        prog.failed = prog.failed || !ok;
      }
      prog.cv.notify_all();
    }
  };
  rsx::HelperPool::Handle up = ctx->helpers.submit(upload);
  if (!up)
    upload();
  int rc = RSX_OK;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    plan->last_stream = s;
    plan->ran = true;
  }
  rc = ljpeg_plan_run_begin(lp, d_in, out_row0, s);
  uint64_t done = 0;
  std::vector<LjRegion> reg;
  auto fetch = [&](uint64_t lo, uint64_t hi) -> int {
    ljpeg_plan_region(lp, lo, hi, &reg);
    for (const LjRegion& g : reg) {
      // (inside the rectangle the job owns; one rectangle a call of download_rects: each is a
      // rectangle by itself, not a ragged set)
      const size_t r0 = std::max(g.row0, whole.row0), r1 = std::min(g.row0 + g.rows, whole.row0 + whole.rows);
      const size_t b0 = std::max(g.byte0, whole.byte0), b1 = std::min(g.byte0 + g.bytes, whole.byte0 + whole.bytes);
      if (r1 <= r0 || b1 <= b0)
        continue;
      const size_t off = r0 * img->pitch_bytes + b0;
      const DownRect dr{static_cast<uint8_t*>(img->data) + off, img->pitch_bytes, out_row0 + off,
                        img->pitch_bytes, b1 - b0, r1 - r0};
      std::lock_guard<std::mutex> down(ctx->download_mu);
      if (int e = download_rects(ctx, L, s, &dr, 1))
        return e;
    }
    return RSX_OK;
  };
  bool up_failed = false;
  for (int c = 0; c < NCH && rc == RSX_OK; ++c) {
    {
      std::unique_lock<std::mutex> g(prog.m);
      prog.cv.wait(g, [&]() { return prog.ready > c; });
      up_failed = prog.failed;
    }
    if (up_failed)
      break;
    if (hipStreamWaitEvent(s, L->ev_up[c], 0) != hipSuccess) {
      rc = RSX_ERR_DEVICE;
      break;
    }
    rc = ljpeg_plan_run_blocks(lp, s, c ? blk_end[c - 1] : 0u, blk_end[c]);
    if (rc == RSX_OK && c + 1 < NCH) {
      uint64_t now = 0;
      rc = ljpeg_plan_symbols_done(lp, s, blk_end[c], &now);
      if (rc == RSX_OK && now > done) {
        rc = fetch(done, now);
        done = now;
      }
    }
  }
  ctx->helpers.wait(up);
  // (both streams drained on every way out: the caller's buffers may go the moment we return)
  const hipError_t e_up = hipStreamSynchronize(s_up);
  if (up_failed || e_up != hipSuccess || rc == RSX_ERR_DEVICE) {
    (void)hipStreamSynchronize(s);
    set_error(ctx, "ljpeg (chunked): upload or launch failed");
    return RSX_ERR_DEVICE;
  }
  if (rc == RSX_OK)
    rc = ljpeg_plan_run_end(lp, s);
  if (rc == RSX_OK)
    rc = rsx_plan_results(plan, status, consumed);
  if (rc == RSX_ERR_DEVICE || rc == RSX_ERR_NOMEM) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  if (*status == RSX_OK) {
    if (ljpeg_plan_single_pass_held(lp)) {
      if (int e = fetch(done, ~uint64_t(0)))
        return e;
    } else {
      // (the stream went through a second pass: what came down early may be stale)
      const size_t off = whole.row0 * img->pitch_bytes + whole.byte0;
      const DownRect dr{static_cast<uint8_t*>(img->data) + off, img->pitch_bytes, out_row0 + off,
                        img->pitch_bytes, whole.bytes, whole.rows};
      std::lock_guard<std::mutex> down(ctx->download_mu);
      if (int e = download_rects(ctx, L, s, &dr, 1))
        return e;
    }
  }
  return rc;
}

template <typename JobT, typename CreateFn>
int ljpeg_family_host(rsx_ctx* ctx, int n, std::vector<JobT>& jobs,
                      const uint8_t* const* ins, const rsx_image* img,
                      CreateFn create, int32_t* statuses, uint32_t* consumed,
                      bool count_call = true) {
  if (count_call)
    ++ctx->host_calls;
#ifdef RSX_FORCE_UNSUPPORTED
  // (experiment build: every LJPEG-family host-pointer call refuses -- the drop-in tests
  // must notice that the images then come from the reference's own loops)
  for (int i = 0; i < n; ++i)
    statuses[i] = RSX_ERR_UNSUPPORTED;
  return RSX_ERR_UNSUPPORTED;
#endif
  // (the image view sizes the staging: check it before anything is allocated or copied;
  // every decompressor's own validation rejects such an image as well)
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->pitch_bytes == 0 || n < 1)
    return RSX_ERR_INVALID_ARG;
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // inputs back to back (16-byte aligned) + 64 zero bytes of slack each
  size_t in_total = 0;
  for (int i = 0; i < n; ++i) {
    jobs[i].in_offset = in_total;
    in_total += align_up(size_t(jobs[i].in_bytes) + 64, 16);
    jobs[i].img = *img;
    jobs[i].img_offset = 0;
  }
  // the output staging holds the image ROWS the jobs write (a DNG tile call: the tile's
  // rows, not the image); the kernels get the address row 0 would have
  size_t row_lo = size_t(img->dim_y), row_hi = 0;
  for (int i = 0; i < n; ++i) {
    const HostRect r = out_rect(jobs[i]);
    row_lo = std::min(row_lo, r.row0);
    row_hi = std::max(row_hi, std::min(r.row0 + r.rows, size_t(img->dim_y)));
  }
  if (row_hi <= row_lo) {
    row_lo = 0;
    row_hi = size_t(img->dim_y);
  }
  const size_t out_bytes = size_t(img->pitch_bytes) * (row_hi - row_lo);
  const size_t out_skip = size_t(img->pitch_bytes) * row_lo;
  // the lane's cached plan, if it was made from these very jobs (descriptors, sizes,
  // offsets, image geometry; not the host pointer of the image)
  std::vector<uint8_t> key;
  key.reserve(sizeof(void*) + size_t(n) * sizeof(JobT));
  {
    const void* fn = reinterpret_cast<const void*>(create);
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&fn);
    key.insert(key.end(), p, p + sizeof fn);
    for (int i = 0; i < n; ++i)
      key_job(key, jobs[i]);
  }
  LaneGuard lane(ctx, &key); // staging + stream of this call
  if (!lane.lane)
    return RSX_ERR_DEVICE;
  if (int e = lane.lane->d_in.ensure(in_total + 64))
    return e;
  if (int e = lane.lane->d_out.ensure(out_bytes + 64))
    return e;
  hipStream_t s = lane.lane->stream;
  // ONE large stream of the single-pass kernel whose plan the lane holds (a frame decoded
  // again: a burst, a folder of one camera's files): in chunks -- see ljpeg_chunked_host.
  if (n == 1 && ctx->host_overlap && jobs[0].in_bytes >= (size_t(8) << 20) &&
      lane.lane->cached_plan && lane.lane->cached_key == key &&
      lane.lane->cached_plan->kind == PLAN_LJPEG && !lane.lane->cached_plan->timing &&
      ljpeg_plan_chunkable(lane.lane->cached_plan->ljpeg.get())) {
    int32_t st1 = RSX_OK;
    uint32_t cons1 = 0;
    const int rc = ljpeg_chunked_host(ctx, lane.lane, lane.lane->cached_plan, jobs[0].in_bytes, in_total,
                                      ins[0], img, out_rect(jobs[0]), out_skip, &st1, &cons1);
    if (rc != CHUNKED_NOT_TAKEN) {
      if (rc == RSX_ERR_DEVICE || rc == RSX_ERR_NOMEM) {
        rsx_plan_destroy(lane.lane->cached_plan);
        lane.lane->cached_plan = nullptr;
        return rc;
      }
      if (statuses)
        statuses[0] = st1;
      if (consumed)
        consumed[0] = cons1;
      return rc;
    }
  }
  {
    // (one upload at a time, one download at a time: calls that run side by side -- the
    // bands of a split DNG call, the files of several threads -- then take turns on each
    // direction of the link instead of sharing both, and one call's download runs under
    // the next one's upload)
    std::lock_guard<std::mutex> up(ctx->upload_mu);
    RSX_HIP_CHECK(ctx, hipMemsetAsync(lane.lane->d_in.ptr, 0, in_total + 64, s));
    for (int i = 0; i < n; ++i)
      RSX_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<uint8_t*>(lane.lane->d_in.ptr) +
                                            jobs[i].in_offset,
                                        ins[i], jobs[i].in_bytes, hipMemcpyHostToDevice, s));
  }
  rsx_plan* plan = nullptr;
  if (lane.lane->cached_plan && lane.lane->cached_key == key) {
    plan = lane.lane->cached_plan;
  } else {
    if (lane.lane->cached_plan)
      rsx_plan_destroy(lane.lane->cached_plan);
    lane.lane->cached_plan = nullptr;
    if (int st = create(ctx, n, jobs.data(), &plan))
      return st;
    lane.lane->cached_plan = plan;
    lane.lane->cached_key = std::move(key);
  }
  std::vector<int32_t> st(n, RSX_OK);
  std::vector<uint32_t> cons(n, 0);
  uint8_t* const out_row0 = static_cast<uint8_t*>(lane.lane->d_out.ptr) - out_skip;
  int rc = rsx_plan_run(plan, lane.lane->d_in.ptr, out_row0, s);
  if (rc == RSX_OK)
    rc = rsx_plan_results(plan, st.data(), cons.data());
  if (rc == RSX_ERR_DEVICE || rc == RSX_ERR_NOMEM) {
    rsx_plan_destroy(lane.lane->cached_plan);
    lane.lane->cached_plan = nullptr;
    return rc;
  }
#ifdef RSX_DIAG_VERIFY
  // (diagnostic build, round 6: an intermittent wrong tile with status OK under six host threads --
  // was the INPUT on the device what the caller handed over, and does the same plan on the same
  // device input give the same pixels a second time?)
  {
    static std::atomic<unsigned long long> calls{0}, bad_in{0}, bad_rerun{0};
    ++calls;
    std::vector<uint8_t> back(in_total + 64), out1(out_bytes), out2(out_bytes);
    (void)hipMemcpyAsync(back.data(), lane.lane->d_in.ptr, in_total + 64, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(out1.data(), lane.lane->d_out.ptr, out_bytes, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    for (int i = 0; i < n; ++i) {
      const uint8_t* h = static_cast<const uint8_t*>(ins[i]);
      const uint8_t* d = back.data() + jobs[i].in_offset;
      size_t nd = 0, first = 0, zeros = 0;
      for (size_t k = 0; k < size_t(jobs[i].in_bytes); ++k)
        if (h[k] != d[k]) {
          if (!nd)
            first = k;
          ++nd;
          zeros += d[k] == 0;
        }
      if (nd) {
        ++bad_in;
        fprintf(stderr, "RSX_DIAG_VERIFY: INPUT of job %d/%d on the device differs from the caller's in %zu of %zu "
                        "bytes (first at %zu; %zu of them zero on the device), in_offset %zu\n",
                i, n, nd, size_t(jobs[i].in_bytes), first, zeros, size_t(jobs[i].in_offset));
      }
    }
    std::vector<int32_t> st2(n, RSX_OK);
    std::vector<uint32_t> cons2(n, 0);
    int rc2 = rsx_plan_run(plan, lane.lane->d_in.ptr, out_row0, s);
    if (rc2 == RSX_OK)
      rc2 = rsx_plan_results(plan, st2.data(), cons2.data());
    (void)hipMemcpyAsync(out2.data(), lane.lane->d_out.ptr, out_bytes, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    size_t nd = 0, first = 0;
    for (size_t k = 0; k < out_bytes; ++k)
      if (out1[k] != out2[k]) {
        if (!nd)
          first = k;
        ++nd;
      }
    if (nd || st2 != st || cons2 != cons || rc2 != rc) {
      ++bad_rerun;
      fprintf(stderr, "RSX_DIAG_VERIFY: a SECOND run of the plan on the same device input differs: %zu of %zu output "
                      "bytes (first at %zu = row %zu byte %zu), rc %d -> %d;",
              nd, out_bytes, first, first / size_t(img->pitch_bytes), first % size_t(img->pitch_bytes), rc, rc2);
      for (int i = 0; i < n; ++i)
        fprintf(stderr, " job %d: status %d -> %d consumed %u -> %u (in_bytes %zu);", i, st[i], st2[i], cons[i],
                cons2[i], size_t(jobs[i].in_bytes));
      fprintf(stderr, " (the second run's results are the ones delivered)\n");
      st = st2;
      cons = cons2;
      rc = rc2;
    }
    if ((calls & 1023ull) == 0)
      fprintf(stderr, "RSX_DIAG_VERIFY: %llu calls, %llu with a device input that differs, %llu with a second run that differs\n",
              (unsigned long long)calls, (unsigned long long)bad_in, (unsigned long long)bad_rerun);
  }
#endif
  // only the rectangle a successful job decoded goes back to the host image:
  // pixels outside it (other tiles, padding) are never touched
  std::vector<HostRect> rects;
  for (int i = 0; i < n; ++i) {
    if (statuses)
      statuses[i] = st[i];
    if (consumed)
      consumed[i] = cons[i];
    if (st[i] == RSX_OK)
      rects.push_back(out_rect(jobs[i]));
  }
  // Tiles that together fill their bounding box (the rule: every tile of a DNG decoded)
  // go back as ONE rectangle -- a pageable 2D copy per tile cost 2 ms more on the four
  // tiles of an 8192x5464 frame than the copy of the whole frame.
  if (rects.size() > 1) {
    size_t r0 = ~size_t(0), r1 = 0, b0 = ~size_t(0), b1 = 0, area = 0;
    for (const HostRect& r : rects) {
      r0 = std::min(r0, r.row0);
      r1 = std::max(r1, r.row0 + r.rows);
      b0 = std::min(b0, r.byte0);
      b1 = std::max(b1, r.byte0 + r.bytes);
      area += r.rows * r.bytes;
    }
    bool disjoint = true; // (tiles of one image never overlap; checked for the few there are)
    for (size_t x = 0; x < rects.size() && disjoint && rects.size() <= 64; ++x)
      for (size_t y = x + 1; y < rects.size(); ++y) {
        const HostRect &p = rects[x], &q = rects[y];
        if (p.row0 < q.row0 + q.rows && q.row0 < p.row0 + p.rows &&
            p.byte0 < q.byte0 + q.bytes && q.byte0 < p.byte0 + p.bytes)
          disjoint = false;
      }
    if (disjoint && rects.size() <= 64 && area == (r1 - r0) * (b1 - b0))
      rects.assign(1, HostRect{r0, r1 - r0, b0, b1 - b0});
  }
  // Back into the caller's image: rectangles on the 16-byte grid straight, everything ragged
  // (3-sample pixels, odd widths, tiles of unequal heights) through the lane's page-locked
  // staging -- download_rects; until round 5 every rectangle was a 2-D copy of the runtime
  // into the pageable image, and one such copy of a rectangle on 2-byte boundaries left
  // sixteen bytes undelivered.
  {
    std::vector<DownRect> down;
    down.reserve(rects.size());
    for (const HostRect& r : rects) {
      const size_t off = r.row0 * img->pitch_bytes + r.byte0;
      down.push_back({static_cast<uint8_t*>(img->data) + off, img->pitch_bytes, out_row0 + off,
                      img->pitch_bytes, r.bytes, r.rows});
    }
    std::lock_guard<std::mutex> down_lock(ctx->download_mu);
#ifdef RSX_DIAG_DOWNLOAD
    // (diagnostic build: the round-5 copies -- one 2-D copy a rectangle into the pageable image)
    for (const DownRect& r : down)
      RSX_HIP_CHECK(ctx, hipMemcpy2DAsync(r.host, r.host_pitch, r.dev, r.dev_pitch, r.bytes, r.rows,
                                          hipMemcpyDeviceToHost, s));
    RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
#else
    if (int e = download_rects(ctx, lane.lane, s, down.data(), down.size()))
      return e;
#endif
  }
#ifdef RSX_DIAG_DOWNLOAD
  {
    // Diagnostic build (scripts/r06a.sh): the same device rows once more, as ONE contiguous
    // copy into memory of our own, and every rectangle of the host image compared with them.
    // A difference = bytes the 2-D copy did not deliver although the device held them.
    static std::atomic<uint64_t> calls{0}, bad_calls{0};
    size_t r0 = ~size_t(0), r1 = 0;
    for (const HostRect& r : rects) {
      r0 = std::min(r0, r.row0);
      r1 = std::max(r1, r.row0 + r.rows);
    }
    const size_t pitch = img->pitch_bytes;
    if (!rects.empty()) {
      std::vector<uint8_t> chk((r1 - r0) * pitch);
      RSX_HIP_CHECK(ctx, hipMemcpy(chk.data(), out_row0 + r0 * pitch, chk.size(), hipMemcpyDeviceToHost));
      ++calls;
      bool bad = false;
      for (size_t i = 0; i < rects.size(); ++i) {
        const HostRect& r = rects[i];
        for (size_t y = 0; y < r.rows; ++y) {
          const uint8_t* h = static_cast<uint8_t*>(img->data) + (r.row0 + y) * pitch + r.byte0;
          const uint8_t* d = chk.data() + (r.row0 + y - r0) * pitch + r.byte0;
          if (std::memcmp(h, d, r.bytes) != 0) {
            size_t a = 0, b = r.bytes;
            while (a < r.bytes && h[a] == d[a]) ++a;
            while (b > a && h[b - 1] == d[b - 1]) --b;
            fprintf(stderr, "RSX_DIAG_DOWNLOAD: rect %zu/%zu (row0 %zu rows %zu byte0 %zu bytes %zu) row %zu: "
                    "host != device in rect-bytes [%zu, %zu) (unit %zu + %zu); host %02x %02x .. device %02x %02x; "
                    "host row addr %p, image %p pitch %zu\n", i, rects.size(), r.row0, r.rows, r.byte0, r.bytes,
                    y, a, b, a / 16, a % 16, h[a], h[a + 1], d[a], d[a + 1], static_cast<const void*>(h),
                    img->data, pitch);
            bad = true;
          }
        }
      }
      if (bad) {
        ++bad_calls;
        auto still_bad = [&](std::vector<uintptr_t>* pages) {
          size_t still = 0;
          for (const HostRect& r : rects)
            for (size_t y = 0; y < r.rows; ++y) {
              const uint8_t* h = static_cast<uint8_t*>(img->data) + (r.row0 + y) * pitch + r.byte0;
              const uint8_t* d = chk.data() + (r.row0 + y - r0) * pitch + r.byte0;
              if (std::memcmp(h, d, r.bytes) != 0) {
                ++still;
                if (pages)
                  for (size_t x = 0; x < r.bytes; ++x)
                    if (h[x] != d[x]) {
                      const uintptr_t pg = reinterpret_cast<uintptr_t>(h + x) & ~uintptr_t(4095);
                      if (pages->empty() || pages->back() != pg)
                        pages->push_back(pg);
                    }
              }
            }
          return still;
        };
        auto again2d = [&]() -> int {
          for (const HostRect& r : rects) {
            const size_t off = r.row0 * pitch + r.byte0;
            RSX_HIP_CHECK(ctx, hipMemcpy2DAsync(static_cast<uint8_t*>(img->data) + off, pitch, out_row0 + off,
                                                pitch, r.bytes, r.rows, hipMemcpyDeviceToHost, s));
          }
          RSX_HIP_CHECK(ctx, hipStreamSynchronize(s));
          return RSX_OK;
        };
        std::vector<uintptr_t> pages;
        still_bad(&pages);
        std::sort(pages.begin(), pages.end());
        pages.erase(std::unique(pages.begin(), pages.end()), pages.end());
        fprintf(stderr, "RSX_DIAG_DOWNLOAD: %zu host pages hold undelivered bytes:", pages.size());
        for (size_t k = 0; k < pages.size() && k < 12; ++k)
          fprintf(stderr, " %p", reinterpret_cast<void*>(pages[k]));
        fprintf(stderr, "\n");
        // /proc/self/pagemap of the first such page, its neighbours and the image's first page
        // (bit 63 present, 62 swapped, 61 file/shared, 56 exclusively mapped, 55 soft-dirty)
        if (FILE* pm = fopen("/proc/self/pagemap", "rb")) {
          auto entry = [&](uintptr_t va) {
            uint64_t e = 0;
            if (fseek(pm, long(va / 4096 * 8), SEEK_SET) == 0 && fread(&e, 8, 1, pm) == 1)
              return e;
            return ~uint64_t(0);
          };
          const uintptr_t pg = pages.empty() ? 0 : pages[0];
          fprintf(stderr, "RSX_DIAG_DOWNLOAD: pagemap bad %016llx prev %016llx next-good %016llx image[0] %016llx\n",
                  (unsigned long long)entry(pg), (unsigned long long)entry(pg - 4096),
                  (unsigned long long)entry(pages.empty() ? 0 : pages.back() + 4096),
                  (unsigned long long)entry(reinterpret_cast<uintptr_t>(img->data)));
          fclose(pm);
        }
        // A0: what does the DEVICE read from those host pages?  The rows up (the runtime pins the
        // caller's pages for that as well), down again into memory of our own, compared with
        // what the CPU reads there: bytes that come back as the decoded pixels although the CPU
        // sees the fill = the device and the CPU look at different physical pages
        {
          void* d_tmp = nullptr;
          if (hipMalloc(&d_tmp, chk.size()) == hipSuccess) {
            std::vector<uint8_t> back(chk.size());
            const uint8_t* rows0 = static_cast<uint8_t*>(img->data) + r0 * pitch;
            RSX_HIP_CHECK(ctx, hipMemcpy(d_tmp, rows0, chk.size(), hipMemcpyHostToDevice));
            RSX_HIP_CHECK(ctx, hipMemcpy(back.data(), d_tmp, chk.size(), hipMemcpyDeviceToHost));
            size_t n_bad = 0, gpu_sees_pixels = 0, gpu_sees_fill = 0;
            for (const HostRect& r : rects)
              for (size_t y = 0; y < r.rows; ++y)
                for (size_t x = 0; x < r.bytes; ++x) {
                  const size_t o = (r.row0 + y - r0) * pitch + r.byte0 + x;
                  if (rows0[o] != chk[o]) {
                    ++n_bad;
                    gpu_sees_pixels += back[o] == chk[o];
                    gpu_sees_fill += back[o] == rows0[o];
                  }
                }
            fprintf(stderr, "RSX_DIAG_DOWNLOAD: A0 of %zu undelivered bytes the device reads %zu as the decoded "
                    "pixels and %zu as what the CPU sees\n", n_bad, gpu_sees_pixels, gpu_sees_fill);
            (void)hipFree(d_tmp);
          }
        }
        // A: the same 2-D copies once more
        if (int e = again2d()) return e;
        fprintf(stderr, "RSX_DIAG_DOWNLOAD: A after repeating the 2-D copies %zu rows still differ\n", still_bad(nullptr));
        // B: the CPU writes one byte of every such page (the value it holds), then the copies again
        for (uintptr_t pg : pages) {
          volatile uint8_t* q = reinterpret_cast<volatile uint8_t*>(pg);
          const uint8_t v = q[0];
          q[0] = v;
        }
        if (int e = again2d()) return e;
        fprintf(stderr, "RSX_DIAG_DOWNLOAD: B after a CPU write to each of the pages + the 2-D copies %zu rows still differ\n", still_bad(nullptr));
        // C: page-lock the rows (hipHostRegister), the copies again
        {
          uint8_t* base = static_cast<uint8_t*>(img->data) + r0 * pitch;
          const hipError_t er = hipHostRegister(base, (r1 - r0) * pitch, hipHostRegisterDefault);
          if (er == hipSuccess) {
            if (int e = again2d()) return e;
            fprintf(stderr, "RSX_DIAG_DOWNLOAD: C with the rows registered (page-locked) %zu rows still differ\n", still_bad(nullptr));
            (void)hipHostUnregister(base);
          } else {
            (void)hipGetLastError();
            fprintf(stderr, "RSX_DIAG_DOWNLOAD: C hipHostRegister failed: %s\n", hipGetErrorString(er));
          }
        }
        // D: ONE contiguous copy of the rows straight into the image (diagnostic only: it writes the padding too)
        RSX_HIP_CHECK(ctx, hipMemcpy(static_cast<uint8_t*>(img->data) + r0 * pitch, out_row0 + r0 * pitch,
                                     (r1 - r0) * pitch, hipMemcpyDeviceToHost));
        fprintf(stderr, "RSX_DIAG_DOWNLOAD: D after one contiguous copy of the rows into the image %zu rows still differ\n", still_bad(nullptr));
      }
      if ((calls & 63) == 0 || bad)
        fprintf(stderr, "RSX_DIAG_DOWNLOAD: %llu calls checked, %llu with undelivered bytes\n",
                (unsigned long long)calls.load(), (unsigned long long)bad_calls.load());
    }
  }
#endif
  return rc;
}

} // namespace

extern "C" int rsx_ljpeg_decode(rsx_ctx* ctx, const rsx_ljpeg_desc* d,
                                const uint8_t* in, size_t in_bytes,
                                const rsx_image* img, uint32_t* consumed) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_ljpeg_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  uint32_t c = 0;
  const int rc = ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_ljpeg_plan_create,
                                   &st, &c);
  if (consumed)
    *consumed = c;
  return rc;
}

extern "C" int rsx_cr2_decode(rsx_ctx* ctx, const rsx_cr2_desc* d,
                              const uint8_t* in, size_t in_bytes,
                              const rsx_image* img, uint32_t* consumed) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_cr2_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  uint32_t c = 0;
  const int rc = ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_cr2_plan_create,
                                   &st, &c);
  if (consumed)
    *consumed = c;
  return rc;
}

extern "C" int rsx_nikon_decompress(rsx_ctx* ctx, const rsx_nikon_desc* d,
                                    const uint8_t* in, size_t in_bytes,
                                    const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_nikon_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  return ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_nikon_plan_create, &st, nullptr);
}

extern "C" int rsx_pentax_decompress(rsx_ctx* ctx, const rsx_pentax_desc* d,
                                     const uint8_t* in, size_t in_bytes,
                                     const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_pentax_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  return ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_pentax_plan_create, &st, nullptr);
}

extern "C" int rsx_hasselblad_decompress(rsx_ctx* ctx, const rsx_hasselblad_desc* d,
                                         const uint8_t* in, size_t in_bytes,
                                         const rsx_image* img, uint32_t* consumed) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_hasselblad_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  uint32_t c = 0;
  const int rc =
      ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_hasselblad_plan_create, &st, &c);
  if (consumed)
    *consumed = c;
  return rc;
}

extern "C" int rsx_samsung_v1_decompress(rsx_ctx* ctx, const rsx_samsung_v1_desc* d,
                                         const uint8_t* in, size_t in_bytes,
                                         const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_samsung_v1_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  return ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_samsung_v1_plan_create, &st, nullptr);
}

extern "C" int rsx_samsung_v2_decompress(rsx_ctx* ctx, const rsx_samsung_v2_desc* d,
                                         const uint8_t* in, size_t in_bytes,
                                         const rsx_image* img) {
  if (!ctx || !d || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_samsung_v2_job> jobs(1);
  jobs[0].desc = *d;
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  return ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_samsung_v2_plan_create, &st, nullptr);
}

extern "C" int rsx_sony_arw1_decompress(rsx_ctx* ctx, const uint8_t* in, size_t in_bytes,
                                        const rsx_image* img) {
  if (!ctx || !in || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_sony_arw1_job> jobs(1);
  jobs[0].in_bytes = in_bytes;
  int32_t st = RSX_OK;
  return ljpeg_family_host(ctx, 1, jobs, &in, img, rsx_sony_arw1_plan_create, &st, nullptr);
}

extern "C" int rsx_dng_decompress_ljpeg(rsx_ctx* ctx, int n_tiles,
                                        const rsx_dng_ljpeg_tile* tiles,
                                        const rsx_image* img,
                                        int32_t* tile_status,
                                        uint32_t* tile_consumed) {
  if (!ctx || !tiles || n_tiles < 1 || !img || !img->data)
    return RSX_ERR_INVALID_ARG;
  std::vector<rsx_ljpeg_job> jobs(n_tiles);
  std::vector<const uint8_t*> ins(n_tiles);
  for (int i = 0; i < n_tiles; ++i) {
    jobs[i].desc = tiles[i].desc;
    jobs[i].in_bytes = tiles[i].in_bytes;
    ins[i] = tiles[i].in;
  }
  constexpr int32_t ST_UNSET = INT32_MIN; // "ljpeg_family_host never got to this tile"
  std::vector<int32_t> st(n_tiles, RSX_OK);
  std::vector<uint32_t> cons(n_tiles, 0);
  // A large image: bands of its tile ROWS (up to four) as calls of their own, side by side on
  // helper threads -- the link is full duplex, and a copy from or to pageable memory keeps
  // its calling thread: while one band is downloading its pixels the next one uploads its
  // bytes and decodes (the calls take turns on each direction, ljpeg_family_host).  cfg 4:
  // 44 MB up, 0.2 ms of kernels, 89 MB down, one after the other 2.74 ms; as two bands 2.40.
  // By tile rows, so that each band's pixels go back as one rectangle.
  int rc = RSX_OK;
  {
    size_t total_in = 0;
    for (int i = 0; i < n_tiles; ++i)
      total_in += tiles[i].in_bytes;
    std::vector<int> band_of(n_tiles, 0);
    int n_bands = 1;
    if (ctx->host_overlap && n_tiles >= 2 && total_in >= (size_t(8) << 20)) {
      std::vector<int32_t> ys;
      for (int i = 0; i < n_tiles; ++i)
        ys.push_back(tiles[i].desc.tile_y);
      std::sort(ys.begin(), ys.end());
      ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
      n_bands = int(std::min<size_t>(4, ys.size()));
      for (int i = 0; i < n_tiles; ++i) {
        const size_t row = size_t(std::lower_bound(ys.begin(), ys.end(), tiles[i].desc.tile_y) -
                                  ys.begin());
        band_of[i] = int(row * size_t(n_bands) / ys.size());
      }
    }
    if (n_bands >= 2) {
      struct Band {
        std::vector<rsx_ljpeg_job> jobs;
        std::vector<const uint8_t*> ins;
        std::vector<int> tile;
        std::vector<int32_t> st;
        std::vector<uint32_t> cons;
        int rc = RSX_OK;
      };
      std::vector<Band> bands(n_bands);
      for (int i = 0; i < n_tiles; ++i) {
        Band& bd = bands[band_of[i]];
        bd.jobs.push_back(jobs[i]);
        bd.ins.push_back(ins[i]);
        bd.tile.push_back(i);
      }
      auto run_band = [&](Band& bd, bool count) {
        bd.st.assign(bd.jobs.size(), ST_UNSET);
        bd.cons.assign(bd.jobs.size(), 0);
        bd.rc = ljpeg_family_host(ctx, int(bd.jobs.size()), bd.jobs, bd.ins.data(), img,
                                  rsx_ljpeg_plan_create, bd.st.data(), bd.cons.data(), count);
      };
      // (persistent helpers of the context; none to be had: in turn, on this thread)
      std::vector<rsx::HelperPool::Handle> helpers;
      std::vector<int> inline_bands;
      for (int k = 1; k < n_bands; ++k) {
        rsx::HelperPool::Handle h = ctx->helpers.submit([&, k]() { run_band(bands[k], false); });
        if (h)
          helpers.push_back(h);
        else
          inline_bands.push_back(k);
      }
      run_band(bands[0], true);
      for (int k : inline_bands)
        run_band(bands[k], false);
      for (const rsx::HelperPool::Handle& h : helpers)
        ctx->helpers.wait(h);
      for (const Band& bd : bands) {
        for (size_t k = 0; k < bd.tile.size(); ++k) {
          // (a band that came back early -- no plan, a bad argument -- never filled its
          // statuses in: its tiles were NOT decoded, whatever the initial value says)
          st[bd.tile[k]] = bd.st[k] == ST_UNSET ? (bd.rc != RSX_OK ? bd.rc : RSX_ERR_DEVICE) : bd.st[k];
          cons[bd.tile[k]] = bd.cons[k];
        }
        if (bd.rc == RSX_ERR_DEVICE || bd.rc == RSX_ERR_NOMEM)
          rc = bd.rc; // (only a device / memory failure of a band matters below)
      }
    } else {
      std::fill(st.begin(), st.end(), ST_UNSET);
      rc = ljpeg_family_host(ctx, n_tiles, jobs, ins.data(), img, rsx_ljpeg_plan_create,
                             st.data(), cons.data());
      for (int32_t& v : st)
        if (v == ST_UNSET)
          v = rc != RSX_OK ? rc : RSX_ERR_DEVICE;
    }
  }
  if (tile_status)
    std::copy(st.begin(), st.end(), tile_status);
  if (tile_consumed)
    std::copy(cons.begin(), cons.end(), tile_consumed);
  if (rc == RSX_ERR_DEVICE || rc == RSX_ERR_NOMEM)
    return rc;
  for (int i = 0; i < n_tiles; ++i)
    if (st[i] != RSX_OK)
      return RSX_ERR_TILE_ERRORS; // AbstractDngDecompressor.cpp:247-251
  return RSX_OK;
}

extern "C" int rsx_probe_stream_copy(rsx_ctx* ctx, const void* in_dev, size_t in_bytes,
                                     void* out_dev, size_t out_bytes, void* stream,
                                     int reps, double* avg_ms) {
  if (!ctx || !in_dev || !out_dev || reps < 1 || !avg_ms)
    return RSX_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  RSX_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  RSX_HIP_CHECK(ctx, hipEventCreate(&e0));
  RSX_HIP_CHECK(ctx, hipEventCreate(&e1));
  hipError_t err = hipEventRecord(e0, s);
  for (int i = 0; i < reps && err == hipSuccess; ++i)
    err = launch_stream_probe(in_dev, in_bytes, out_dev, out_bytes, s);
  if (err == hipSuccess)
    err = hipEventRecord(e1, s);
  if (err == hipSuccess)
    err = hipEventSynchronize(e1);
  float ms = 0.f;
  if (err == hipSuccess)
    err = hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  RSX_HIP_CHECK(ctx, err);
  *avg_ms = double(ms) / reps;
  return RSX_OK;
}
