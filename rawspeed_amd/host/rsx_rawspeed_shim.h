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

#include "adt/Array1DRef.h"
#include "adt/Point.h"
#include "codes/PrefixCodeDecoder.h"
#include "common/RawImage.h"
#include "decoders/RawDecoderException.h"
#include "io/IOException.h"

#include <cstdint>
#include <mutex>
#include <vector>

namespace rawspeed::rsx_shim {

// One lazily created context per process; rsx calls are re-entrant per context
// (the DNG tile threads may all enter: AbstractDngDecompressor.cpp:112-131).
inline rsx_ctx* context() {
  static rsx_ctx* ctx = [] {
    rsx_ctx* c = nullptr;
    if (rsx_ctx_create(/*device=*/0, &c) != RSX_OK)
      ThrowRDE("rsx: no usable MI355X device");
    return c;
  }();
  return ctx;
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
