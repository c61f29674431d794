// rsx_rawspeed_shim.h -- the reference-side binding of the C-ABI (include/rsx.h).
//
// This header is what a rawspeed maintainer drops into src/librawspeed/ (see
// INTEGRATION.md): it converts the objects the three hot-path decompressors
// already hold (RawImage, iRectangle2D, PrefixCodeDecoder<> recipes, input
// views) into the plain-C descriptors of include/rsx.h, calls the GPU core and
// turns a non-OK status back into the exception type the reference throws.
// It contains no decoding logic and no copy of reference code; it only reads
// public members of reference types:
//   RawImageData::{dim, pitch, getCpp(), isCFA, getU16DataAsUncroppedArray2DRef()}
//     (common/RawImage.h:104-199, :289-296)
//   AbstractPrefixCodeTranscoder::{code, handleDNGBug16()}
//     (codes/AbstractPrefixCodeTranscoder.h:45,82-84), PrefixCode::nCodesPerLength
//     (codes/PrefixCode.h:45), AbstractPrefixCode::codeValues (codes/AbstractPrefixCode.h:190)
#pragma once

#include "rsx.h"
#include "rsx_pin.h"

#include "adt/Array1DRef.h"
#include "adt/Point.h"
#include "codes/PrefixCodeDecoder.h"
#include "common/RawImage.h"
#include "decoders/RawDecoderException.h"
#include "io/IOException.h"

#include <atomic>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <vector>

namespace rawspeed::rsx_shim {

// (the process-wide context() and the optional page-locked pool: rsx_pin.h)

// What became of the units of work (a strip, a scan, a DNG tile) the hunks saw: decoded
// by the device, or left to the method's original body.  Diagnostics only -- the tests
// use it to show that an image was produced by the GPU and not by a silent fall-through.
struct Stats {
  std::atomic<uint64_t> forwarded{0}, fell_through{0};
};
inline Stats& stats() {
  static Stats s;
  return s;
}
// the test every hunk makes on its call's status
inline bool done(int status) {
  ++(status == RSX_OK ? stats().forwarded : stats().fell_through);
  return status == RSX_OK;
}

inline rsx_image view(const RawImage& img) {
  rsx_image v{};
  const auto a = img->getByteDataAsUncroppedArray2DRef(); // UINT16 and F32 images
  v.data = &a(0, 0);
  v.pitch_bytes = implicit_cast<uint32_t>(img->pitch);
  v.dim_x = img->dim.x;
  v.dim_y = img->dim.y;
  v.cpp = implicit_cast<int32_t>(img->getCpp());
  v.is_cfa = img->isCFA ? 1 : 0;
  return v;
}

// status -> the exception the reference would have thrown
[[noreturn]] inline void raise(int st) {
  switch (st) {
  case RSX_ERR_IO:
    ThrowIOE("rsx: %s (out of bounds / truncated input)", rsx_status_string(st));
  case RSX_ERR_INPUT_OVERFLOW:
    ThrowIOE("Buffer overflow read in BitStreamer (rsx)");
  case RSX_ERR_BAD_HUFFMAN_CODE:
    ThrowRDE("bad Huffman code (rsx)");
  default:
    ThrowRDE("rsx: %s: %s", rsx_status_string(st), rsx_ctx_last_error(context()));
  }
}

// ---------------------------------------------------------------------------
// AbstractDngDecompressor::decompress(): ONE batched call for all tiles
// (INTEGRATION.md 4).  The reference's own fan-out (decompressThread<1> / <7>, an
// `omp for` over the tiles) runs unchanged, but while a batch is registered for the
// tiles' input ranges the per-tile hunks of UncompressedDecompressor /
// LJpegDecompressor RECORD their descriptor instead of decoding.  run() then hands
// all recorded tiles to rsx_dng_decompress_* in one call.  Tiles the device path did
// not finish (damaged stream, unsupported shape, no memory) are left to a second
// pass of the same fan-out in which the hunks of the finished tiles return at once
// and the others fall through to the original CPU code -- which decodes them, or
// throws exactly what the reference throws.
// ---------------------------------------------------------------------------
class DngBatch final {
public:
  struct Slot {
    int kind = 0; // 0 = not recorded, 1 = LJPEG scan, 2 = uncompressed strip
    rsx_dng_ljpeg_tile lj{};
    rsx_dng_unpack_tile up{};
    int32_t status = RSX_ERR_UNSUPPORTED;
    uint32_t consumed = 0;
  };
  struct Found {
    DngBatch* batch = nullptr;
    Slot* slot = nullptr;
  };

  DngBatch() = default;
  DngBatch(const DngBatch&) = delete;
  DngBatch& operator=(const DngBatch&) = delete;
  ~DngBatch() {
    std::unique_lock lock(registry().m);
    for (auto it = registry().ranges.begin(); it != registry().ranges.end();)
      it = it->second.batch == this ? registry().ranges.erase(it) : std::next(it);
  }

  // one tile: the bytes the reference hands to its per-tile decompressor
  void add(const uint8_t* begin, size_t size) {
    slots.emplace_back();
    pending.push_back({begin, size});
  }
  void registerTiles() {
    std::unique_lock lock(registry().m);
    for (size_t i = 0; i < pending.size(); ++i)
      registry().ranges[pending[i].first] = {pending[i].first + pending[i].second, this, &slots[i]};
  }
  // the batch (if any) whose tile holds `p`
  static Found find(const uint8_t* p) {
    std::shared_lock lock(registry().m);
    auto& r = registry().ranges;
    auto it = r.upper_bound(p);
    if (it == r.begin())
      return {};
    --it;
    if (p >= it->second.end)
      return {};
    return {it->second.batch, it->second.slot};
  }

  bool replaying = false;

  // true: every recorded tile is done; false: a second (CPU) pass is needed
  bool run(const rsx_image& img) {
    std::vector<rsx_dng_ljpeg_tile> lj;
    std::vector<rsx_dng_unpack_tile> up;
    std::vector<Slot*> ljs, ups;
    for (Slot& s : slots) {
      if (s.kind == 1) {
        lj.push_back(s.lj);
        ljs.push_back(&s);
      } else if (s.kind == 2) {
        up.push_back(s.up);
        ups.push_back(&s);
      }
    }
    bool all = true;
    if (!lj.empty()) {
      std::vector<int32_t> st(lj.size(), RSX_ERR_DEVICE);
      std::vector<uint32_t> cons(lj.size(), 0);
      (void)rsx_dng_decompress_ljpeg(context(), implicit_cast<int>(lj.size()), lj.data(), &img,
                                     st.data(), cons.data());
      for (size_t i = 0; i < ljs.size(); ++i) {
        ljs[i]->status = st[i];
        // LJpegDecoder's marker walk went on from the end of the scan (endOfScan()).  A
        // tile decoded to its full height must have stopped exactly there.  A
        // bottom-overhanging tile stops earlier; that makes no difference to the walk
        // unless there are restart markers left in between, which the reference trips
        // over.  Anything else is redone by the original code.
        const rsx_ljpeg_desc& d = lj[i].desc;
        const bool dri = d.rows_per_restart_interval < d.frame_h;
        const bool full = d.tile_h >= d.frame_h * d.mcu_h;
        if (st[i] == RSX_OK && cons[i] != ljs[i]->consumed && (full || dri))
          ljs[i]->status = RSX_ERR_UNSUPPORTED;
        all = done(ljs[i]->status) && all;
      }
    }
    if (!up.empty()) {
      std::vector<int32_t> st(up.size(), RSX_ERR_DEVICE);
      (void)rsx_dng_decompress_uncompressed(context(), implicit_cast<int>(up.size()), up.data(),
                                            &img, st.data());
      for (size_t i = 0; i < ups.size(); ++i) {
        ups[i]->status = st[i];
        all = done(st[i]) && all;
      }
    }
    return all;
  }

  // Offset of the marker that ends the entropy-coded segment [p, p + n) -- the first
  // FF xx with xx neither 00 (a stuffed FF) nor D0..D7 (RSTn); n if there is none.
  // A tile that is decoded to its full height ends ON that marker, and in a well-formed
  // tile it is the last thing in the buffer: it is looked for from the end (a few bytes)
  // and run() checks the guess against what the decode reports -- a stream with an
  // earlier marker is redone by the original code.  A bottom-overhanging tile stops
  // before the marker, nothing checks the guess then, so it is searched from the front.
  static uint32_t endOfScan(const uint8_t* p, size_t n, bool full_height) {
    auto is_end = [&](size_t i) {
      // (FF FF is fill, not a marker: AbstractLJpegDecoder's peekMarker does not take it)
      return p[i] == 0xFF && p[i + 1] != 0x00 && p[i + 1] != 0xFF &&
             (p[i + 1] < 0xD0 || p[i + 1] > 0xD7);
    };
    if (full_height) {
      for (size_t i = n; i >= 2; --i)
        if (is_end(i - 2))
          return implicit_cast<uint32_t>(i - 2);
      return implicit_cast<uint32_t>(n);
    }
    const uint8_t* q = p;
    const uint8_t* const last = p + (n ? n - 1 : 0);
    while (q < last) {
      q = static_cast<const uint8_t*>(std::memchr(q, 0xFF, size_t(last - q)));
      if (!q)
        break;
      if (is_end(size_t(q - p)))
        return implicit_cast<uint32_t>(q - p);
      ++q;
    }
    return implicit_cast<uint32_t>(n);
  }

private:
  struct Range {
    const uint8_t* end;
    DngBatch* batch;
    Slot* slot;
  };
  struct Registry {
    std::shared_mutex m;
    std::map<const uint8_t*, Range> ranges;
  };
  static Registry& registry() {
    static Registry r;
    return r;
  }
  std::vector<Slot> slots;
  std::vector<std::pair<const uint8_t*, size_t>> pending;
};

// DHT payload of a borrowed decoder -> rsx_huff_table
inline rsx_huff_table table(const PrefixCodeDecoder<>& ht) {
  rsx_huff_table t{};
  const auto& n = ht.code.nCodesPerLength; // index = code length
  for (size_t l = 1; l < n.size() && l <= 16; ++l)
    t.n_codes_per_length[l - 1] = implicit_cast<uint8_t>(n[l]);
  const auto& v = ht.code.codeValues;
  for (size_t i = 0; i < v.size() && i < RSX_MAX_CODE_VALUES; ++i)
    t.code_values[i] = v[i];
  t.n_code_values = implicit_cast<uint8_t>(v.size());
  t.fix_dng_bug16 = ht.handleDNGBug16() ? 1 : 0;
  return t;
}

// recipes -> (tables[], table_index[], init_pred[]), de-duplicating by address
// the way AbstractLJpegDecoder already de-duplicates by content
// (AbstractLJpegDecoder.cpp:258-263)
template <typename Recipe, typename Desc>
inline void recipes(const std::vector<Recipe>& rec, Desc* d) {
  std::vector<const PrefixCodeDecoder<>*> seen;
  d->n_tables = 0;
  for (size_t c = 0; c < rec.size() && c < RSX_MAX_COMPONENTS; ++c) {
    const PrefixCodeDecoder<>* p = &rec[c].ht;
    size_t k = 0;
    while (k < seen.size() && seen[k] != p)
      ++k;
    if (k == seen.size()) {
      seen.push_back(p);
      d->tables[d->n_tables++] = table(*p);
    }
    d->table_index[c] = implicit_cast<uint8_t>(k);
    d->init_pred[c] = rec[c].initPred;
  }
}

} // namespace rawspeed::rsx_shim
