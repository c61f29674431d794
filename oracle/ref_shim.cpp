// TEST INFRASTRUCTURE -- not part of the product.
//
// C wrapper around the *unmodified* reference implementation (rawspeed),
// compiled from the sources where they lie under /root/reference into
// oracle/_ref/librawspeed_ref.so (see oracle/Makefile).  No reference source
// is copied into this repository; this file is our own glue that constructs
// the reference's objects and calls the reference's own entry points:
//   UncompressedDecompressor::readUncompressedRaw  (UncompressedDecompressor.cpp:202)
//   LJpegDecompressor::decode                      (LJpegDecompressor.cpp:341)
//   Cr2Decompressor<>::decompress                  (Cr2DecompressorImpl.h:471)
//   LJpegDecoder::decode / Cr2LJpegDecoder::decode (container level)
//   AbstractDngDecompressor::decompress            (AbstractDngDecompressor.cpp:240)
//   RawParser::getDecoder + RawDecoder::decodeRaw   (whole files: RawParser.cpp:45, RawDecoder.cpp:320)
// It is used (a) to validate oracle/rsx_oracle.c, (b) to generate/verify
// golden vectors, (c) as bench.py's cpu_baseline (kind "reference").
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.

#include "rawspeedconfig.h"

#include "adt/Array1DRef.h"
#include "adt/Point.h"
#include "bitstreams/BitStreams.h"
#include "codes/HuffmanCode.h"
#include "codes/PrefixCodeDecoder.h"
#include "common/RawImage.h"
#include "common/RawspeedException.h"
#include "decoders/RawDecoder.h"
#include "decoders/RawDecoderException.h"
#include "decompressors/AbstractDngDecompressor.h"
#include "decompressors/Cr2Decompressor.h"
#include "decompressors/Cr2LJpegDecoder.h"
#include "decompressors/HasselbladDecompressor.h"
#include "decompressors/LJpegDecoder.h"
#include "decompressors/LJpegDecompressor.h"
#include "decompressors/NikonDecompressor.h"
#include "decompressors/PentaxDecompressor.h"
#include "decompressors/SamsungV1Decompressor.h"
#include "decompressors/SamsungV2Decompressor.h"
#include "decompressors/SonyArw1Decompressor.h"
#include "decompressors/UncompressedDecompressor.h"
#include "interpolators/Cr2sRawInterpolator.h"
#include "io/Buffer.h"
#include "io/ByteStream.h"
#include "io/Endianness.h"
#include "io/IOException.h"
#include "parsers/RawParser.h"

#include "../include/rsx.h"
#ifdef RSX_PATCHED_BUILD
#include "rsx_rawspeed_shim.h"
#endif

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace rawspeed;

namespace {

thread_local std::string g_last_error;
int g_threads = 1;

struct RefImage {
  RawImage img;
  explicit RefImage(RawImage i) : img(std::move(i)) {}
};

int classify(const std::exception& e) {
  g_last_error = e.what();
  const std::string& s = g_last_error;
  if (s.find("bad Huffman code") != std::string::npos)
    return RSX_ERR_BAD_HUFFMAN_CODE;
  if (s.find("restart marker") != std::string::npos ||
      s.find("Jpeg marker not encountered") != std::string::npos ||
      s.find("Not a restart marker") != std::string::npos)
    return RSX_ERR_RESTART_MARKER;
  if (s.find("Buffer overflow read in BitStreamer") != std::string::npos)
    return RSX_ERR_INPUT_OVERFLOW;
  if (s.find("Too many errors") != std::string::npos)
    return RSX_ERR_TILE_ERRORS;
  if (dynamic_cast<const IOException*>(&e) != nullptr)
    return RSX_ERR_IO;
  return RSX_ERR_INVALID_ARG;
}

PrefixCodeDecoder<> makeTable(const rsx_huff_table& t) {
  HuffmanCode<BaselineCodeTag> hc;
  const Buffer counts(t.n_codes_per_length, 16);
  const auto n = hc.setNCodesPerLength(counts);
  if (n != t.n_code_values)
    ThrowRDE("code value count mismatch");
  hc.setCodeValues(Array1DRef<const uint8_t>(t.code_values, int(n)));
  PrefixCodeDecoder<> d(std::move(hc));
  d.setup(/*fullDecode=*/true, t.fix_dng_bug16 != 0);
  return d;
}

template <typename F> int guarded(F&& f) {
  try {
    g_last_error.clear();
    f();
    return RSX_OK;
  } catch (const RawspeedException& e) {
    return classify(e);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return RSX_ERR_INVALID_ARG;
  }
}

} // namespace

// The embedding application must provide this hook (common/Common.h:41).
extern "C" int rawspeed_get_number_of_processor_cores() { return g_threads; }

extern "C" {

const char* ref_last_error() { return g_last_error.c_str(); }

void ref_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

// the optional page-locked pool behind AlignedAllocator (patched build
// only; -1 where the build has no such thing)
int ref_set_pinned_pool(int on) {
#ifdef RSX_PATCHED_BUILD
  rawspeed::rsx_shim::set_pinned_pool(on != 0);
  return on != 0;
#else
  (void)on;
  return -1;
#endif
}

// The image's ErrorLog (what AbstractDngDecompressor's tile threads append to),
// newline-separated, so that the tests can compare it between the two builds.
int ref_image_errors(void* h, char* out, int cap) {
  auto* r = static_cast<RefImage*>(h);
  std::string all;
  const std::vector<std::string> errs = r->img->getErrors(); // (moves them out of the log)
  for (const auto& e : errs) {
    all += e;
    all += '\n';
  }
  if (out && cap > 0) {
    const int n = int(all.size()) < cap - 1 ? int(all.size()) : cap - 1;
    std::memcpy(out, all.data(), size_t(n));
    out[n] = 0;
  }
  return int(all.size());
}

// Host-pointer calls the C-ABI has served for this build's context (-1: this is the
// unmodified build) -- the batched DNG hunk makes ONE per decompress().
long ref_rsx_host_calls() {
#ifdef RSX_PATCHED_BUILD
  rsx_ctx* c = rawspeed::rsx_shim::context();
  return c ? long(rsx_ctx_host_calls(c)) : 0;
#else
  return -1;
#endif
}

// Units of work (strips, scans, tiles) the patched methods handed to the device and got
// back decoded / left to their original bodies (-1: this is the unmodified build).
long ref_rsx_forwarded() {
#ifdef RSX_PATCHED_BUILD
  return long(rawspeed::rsx_shim::stats().forwarded.load());
#else
  return -1;
#endif
}
long ref_rsx_fell_through() {
#ifdef RSX_PATCHED_BUILD
  return long(rawspeed::rsx_shim::stats().fell_through.load());
#else
  return -1;
#endif
}

int ref_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// RawImage lifetime: UINT16 image of dim_x x dim_y pixels, cpp samples each.
void* ref_image_create(int dim_x, int dim_y, int cpp, int is_cfa) {
  try {
    RawImage img = RawImage::create(RawImageType::UINT16);
    img->dim = iPoint2D(dim_x, dim_y);
    img->setCpp(cpp);
    img->isCFA = is_cfa != 0;
    img->createData();
    return new RefImage(std::move(img));
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return nullptr;
  }
}
void* ref_image_create_f32(int dim_x, int dim_y, int cpp) {
  try {
    RawImage img = RawImage::create(RawImageType::F32);
    img->dim = iPoint2D(dim_x, dim_y);
    img->setCpp(cpp);
    img->createData();
    return new RefImage(std::move(img));
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return nullptr;
  }
}
void ref_image_destroy(void* h) { delete static_cast<RefImage*>(h); }

// A whole file through the reference's front door, the way rstest / darktable call it
// (rstest.cpp:245-262): RawParser::getDecoder() picks the per-camera decoder,
// decodeRaw() runs it.  No camera database (meta == nullptr), no decodeMetaData().
// Returns the decoded image (ref_image_* accessors; the caller destroys it) or nullptr.
void* ref_decode_file(const uint8_t* file, size_t bytes, int uncorrected, int* status) {
  RefImage* out = nullptr;
  const int st = guarded([&] {
    const Buffer buf(file, implicit_cast<Buffer::size_type>(bytes));
    RawParser parser(buf);
    std::unique_ptr<RawDecoder> dec = parser.getDecoder(nullptr);
    dec->failOnUnknown = false;
    dec->interpolateBadPixels = false;
    dec->uncorrectedRawValues = uncorrected != 0;
    RawImage img = dec->decodeRaw();
    out = new RefImage(img);
  });
  if (status)
    *status = st;
  return out;
}
// [0..1] uncropped dim, [2] cpp, [3] pitch, [4..5] cropped dim, [6..7] crop offset,
// [8] 0 = UINT16 / 1 = F32, [9] isCFA
void ref_image_info(void* h, int out[10]) {
  const RawImage& img = static_cast<RefImage*>(h)->img;
  const iPoint2D u = img->getUncroppedDim(), off = img->getCropOffset();
  out[0] = u.x;
  out[1] = u.y;
  out[2] = int(img->getCpp());
  out[3] = img->pitch;
  out[4] = img->dim.x;
  out[5] = img->dim.y;
  out[6] = off.x;
  out[7] = off.y;
  out[8] = img->getDataType() == RawImageType::F32 ? 1 : 0;
  out[9] = img->isCFA ? 1 : 0;
}
// what Cr2Decoder sets for sRaw files before decoding (Cr2Decoder.cpp sRaw path);
// AbstractLJpegDecoder::parseSOF checks the SOF against it (:172-176)
void ref_image_set_subsampling(void* h, int x, int y) {
  static_cast<RefImage*>(h)->img->metadata.subsampling = iPoint2D(x, y);
}
uint8_t* ref_image_data(void* h) {
  auto a = static_cast<RefImage*>(h)->img->getByteDataAsUncroppedArray2DRef();
  return reinterpret_cast<uint8_t*>(&a(0, 0));
}
int ref_image_pitch(void* h) { return static_cast<RefImage*>(h)->img->pitch; }
void ref_image_fill(void* h, int byte) {
  auto* r = static_cast<RefImage*>(h);
  std::memset(ref_image_data(h), byte,
              size_t(r->img->pitch) * size_t(r->img->dim.y));
}

int ref_unpack_u16(void* h, const rsx_unpack_desc* d, const uint8_t* in,
                   size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    UncompressedDecompressor u(
        bs, r->img,
        iRectangle2D({d->crop_x, d->crop_y}, {d->crop_w, d->crop_h}),
        d->input_pitch_bytes, d->bits_per_pixel,
        static_cast<BitOrder>(d->bit_order));
    u.readUncompressedRaw();
  });
}

// The variant entry points, constructed the way their callers do
// (DcsDecoder.cpp:72-79, ErfDecoder.cpp:61-67, OrfDecoder.cpp:213-245,
// Rw2Decoder.cpp:98-111).
int ref_unpack_variant_u16(void* h, const rsx_unpack_variant_desc* d,
                           const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    const int w = d->w;
    const iRectangle2D crop({0, 0}, iPoint2D(w, d->h));
    switch (d->variant) {
    case RSX_UNPACK_8BIT_RAW: {
      UncompressedDecompressor u(bs, r->img, crop, 8 * w / 8, 8, BitOrder::LSB);
      u.decode8BitRaw<true>();
      break;
    }
    case RSX_UNPACK_12BIT_WITH_CONTROL: {
      UncompressedDecompressor u(bs, r->img, crop, (12 * w / 8) + ((w + 2) / 10),
                                 12, d->big_endian ? BitOrder::MSB : BitOrder::LSB);
      if (d->big_endian)
        u.decode12BitRawWithControl<Endianness::big>();
      else
        u.decode12BitRawWithControl<Endianness::little>();
      break;
    }
    case RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED: {
      UncompressedDecompressor u(bs, r->img, crop, 16 * w / 8, 16,
                                 d->big_endian ? BitOrder::MSB : BitOrder::LSB);
      if (d->big_endian)
        u.decode12BitRawUnpackedLeftAligned<Endianness::big>();
      else
        u.decode12BitRawUnpackedLeftAligned<Endianness::little>();
      break;
    }
    default:
      ThrowRDE("unknown variant");
    }
  });
}

// decode8BitRaw<false> under a RawImageCurveGuard, as DcsDecoder.cpp:68-81 does
int ref_decode8bit_lookup(void* h, const uint16_t* curve, int curve_size, int w,
                          int hh, const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    const std::vector<uint16_t> table(curve, curve + curve_size);
    RawImageCurveGuard curveHandler(&r->img, table, /*uncorrectedRawValues=*/false);
    UncompressedDecompressor u(bs, r->img, iRectangle2D({0, 0}, iPoint2D(w, hh)),
                               8 * w / 8, 8, BitOrder::LSB);
    u.decode8BitRaw<false>();
  });
}

int ref_ljpeg_decompress(void* h, const rsx_ljpeg_desc* d, const uint8_t* in,
                         size_t in_bytes, uint32_t* consumed) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    std::vector<PrefixCodeDecoder<>> tables;
    tables.reserve(RSX_MAX_COMPONENTS);
    for (int i = 0; i < d->n_tables; ++i)
      tables.emplace_back(makeTable(d->tables[i]));
    std::vector<LJpegDecompressor::PerComponentRecipe> rec;
    for (int c = 0; c < d->n_comp; ++c) {
      if (d->table_index[c] >= tables.size())
        ThrowRDE("bad table index");
      rec.push_back({tables[d->table_index[c]], d->init_pred[c]});
    }
    const LJpegDecompressor::Frame frame{iPoint2D(d->mcu_w, d->mcu_h),
                                         iPoint2D(d->frame_w, d->frame_h)};
    LJpegDecompressor dec(
        r->img,
        iRectangle2D({d->tile_x, d->tile_y}, {d->tile_w, d->tile_h}), frame,
        rec, d->rows_per_restart_interval,
        Array1DRef<const uint8_t>(in, implicit_cast<int>(in_bytes)));
    const auto n = dec.decode();
    if (consumed)
      *consumed = n;
  });
}

int ref_cr2_decompress(void* h, const rsx_cr2_desc* d, const uint8_t* in,
                       size_t in_bytes, uint32_t* consumed) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    std::vector<PrefixCodeDecoder<>> tables;
    tables.reserve(RSX_MAX_COMPONENTS);
    for (int i = 0; i < d->n_tables; ++i)
      tables.emplace_back(makeTable(d->tables[i]));
    using Dec = Cr2Decompressor<PrefixCodeDecoder<>>;
    std::vector<Dec::PerComponentRecipe> rec;
    for (int c = 0; c < d->n_comp; ++c) {
      if (d->table_index[c] >= tables.size())
        ThrowRDE("bad table index");
      rec.push_back({tables[d->table_index[c]], d->init_pred[c]});
    }
    Dec dec(r->img, std::make_tuple(d->n_comp, d->x_s_f, d->y_s_f),
            iPoint2D(d->frame_w, d->frame_h),
            Cr2SliceWidths(implicit_cast<uint16_t>(d->num_slices),
                           implicit_cast<uint16_t>(d->slice_width),
                           implicit_cast<uint16_t>(d->last_slice_width)),
            rec, Array1DRef<const uint8_t>(in, implicit_cast<int>(in_bytes)));
    const auto n = dec.decompress();
    if (consumed)
      *consumed = n;
  });
}

// Container level: a whole SOI..EOI blob through LJpegDecoder
// (decompressors/LJpegDecoder.cpp:66-102).
int ref_ljpeg_decode_container(void* h, const uint8_t* blob, size_t blob_bytes,
                               uint32_t off_x, uint32_t off_y, uint32_t w,
                               uint32_t hgt, int max_dim_x, int max_dim_y,
                               int fix_dng_bug16) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(blob, implicit_cast<Buffer::size_type>(blob_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    LJpegDecoder d(bs, r->img);
    d.decode(off_x, off_y, w, hgt, iPoint2D(max_dim_x, max_dim_y),
             fix_dng_bug16 != 0);
  });
}

// NikonDecompressor, driven the way NefDecoder does (NefDecoder.cpp:133-135):
// the constructor parses the makernote linearisation blob (`meta`, big-endian
// TIFF entry data).
int ref_nikon_decompress(void* h, const uint8_t* meta, size_t meta_bytes,
                         uint32_t bits_ps, const uint8_t* in, size_t in_bytes,
                         int uncorrected_raw_values) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer mb(meta, implicit_cast<Buffer::size_type>(meta_bytes));
    NikonDecompressor n(r->img, ByteStream(DataBuffer(mb, Endianness::big)), bits_ps);
    n.decompress(Array1DRef<const uint8_t>(in, implicit_cast<int>(in_bytes)),
                 uncorrected_raw_values != 0);
  });
}

// PentaxDecompressor (PefDecoder.cpp): `meta` = the makernote Huffman description
// (big-endian), or NULL for the legacy tree.
int ref_pentax_decompress(void* h, const uint8_t* meta, size_t meta_bytes,
                          const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    Optional<ByteStream> md;
    const Buffer mb(meta, implicit_cast<Buffer::size_type>(meta_bytes));
    if (meta)
      md = ByteStream(DataBuffer(mb, Endianness::big));
    PentaxDecompressor p(r->img, md);
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    p.decompress(ByteStream(DataBuffer(b, Endianness::little)));
  });
}

// SamsungV1Decompressor, as SrwDecoder.cpp:107-119 drives it
int ref_samsung_v1_decompress(void* h, int bits, const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    SamsungV1Decompressor s1(r->img, ByteStream(DataBuffer(b, Endianness::little)), bits);
    s1.decompress();
  });
}

// SonyArw1Decompressor, as ArwDecoder drives it (ArwDecoder.cpp:133-136, :252-254)
// SamsungV2Decompressor (not served by librsx yet; the oracle's restatement is pinned
// against this): the strip as SrwDecoder hands it over, header included.
int ref_samsung_v2_decompress(void* h, int bits, const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    SamsungV2Decompressor d(r->img, bs, implicit_cast<unsigned>(bits));
    d.decompress();
  });
}

int ref_sony_arw1_decompress(void* h, const uint8_t* in, size_t in_bytes) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(in, implicit_cast<Buffer::size_type>(in_bytes));
    SonyArw1Decompressor a(r->img);
    a.decompress(ByteStream(DataBuffer(b, Endianness::little)));
  });
}

// Cr2sRawInterpolator, set up like Cr2Decoder::sRawInterpolate (Cr2Decoder.cpp:585-625):
// `h_in` the subsampled image (cpp 1), `h_out` the interpolated one (cpp 3)
int ref_sraw_interpolate(void* h_in, void* h_out, const rsx_sraw_desc* d) {
  auto* in = static_cast<RefImage*>(h_in);
  auto* out = static_cast<RefImage*>(h_out);
  return guarded([&] {
    out->img->metadata.subsampling = iPoint2D(2, d->subsampling_y);
    out->img->isCFA = false;
    Cr2sRawInterpolator i(out->img, in->img->getU16DataAsUncroppedArray2DRef(),
                          {d->sraw_coeffs[0], d->sraw_coeffs[1], d->sraw_coeffs[2]}, d->hue);
    i.interpolate(d->version);
  });
}

// HasselbladDecompressor, set up like HasselbladLJpegDecoder::decodeScan
// (HasselbladLJpegDecoder.cpp:50-66) with a non-full-decode table
int ref_hasselblad_decompress(void* h, const rsx_hasselblad_desc* d, const uint8_t* in,
                              size_t in_bytes, uint32_t* consumed) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    HuffmanCode<BaselineCodeTag> hc;
    const Buffer nb(d->table.n_codes_per_length, 16);
    const auto n = hc.setNCodesPerLength(nb);
    if (n != d->table.n_code_values)
      ThrowRDE("code value count mismatch");
    hc.setCodeValues(Array1DRef<const uint8_t>(d->table.code_values, implicit_cast<int>(n)));
    PrefixCodeDecoder<> ht(std::move(hc));
    ht.setup(/*fullDecode=*/false, /*fixDNGBug16=*/false);
    const HasselbladDecompressor::PerComponentRecipe rec = {ht, d->init_pred};
    HasselbladDecompressor dec(r->img, rec,
                               Array1DRef<const uint8_t>(in, implicit_cast<int>(in_bytes)));
    const auto c = dec.decompress();
    if (consumed)
      *consumed = c;
  });
}

// Container level: Cr2LJpegDecoder::decode (Cr2LJpegDecoder.cpp:156-165).
int ref_cr2_decode_container(void* h, const uint8_t* blob, size_t blob_bytes,
                             int num_slices, int slice_width,
                             int last_slice_width) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const Buffer b(blob, implicit_cast<Buffer::size_type>(blob_bytes));
    const ByteStream bs(DataBuffer(b, Endianness::little));
    Cr2LJpegDecoder d(bs, r->img);
    d.decode(Cr2SliceWidths(implicit_cast<uint16_t>(num_slices),
                            implicit_cast<uint16_t>(slice_width),
                            implicit_cast<uint16_t>(last_slice_width)));
  });
}

// AbstractDngDecompressor::decompress over `n_tiles` tiles, row-major tile
// order; compression 7 (LJPEG blobs) or 1 (uncompressed, `bps` bits).
// tile_blobs[i] / tile_bytes[i] = data of tile i.
int ref_dng_decompress(void* h, int compression, uint32_t tile_w,
                       uint32_t tile_h, int n_tiles,
                       const uint8_t* const* tile_blobs,
                       const size_t* tile_bytes, int fix_ljpeg, uint32_t bps,
                       int big_endian) {
  auto* r = static_cast<RefImage*>(h);
  return guarded([&] {
    const DngTilingDescription dsc(r->img->dim, tile_w, tile_h);
    if (int(dsc.numTiles) != n_tiles)
      ThrowRDE("tile count mismatch: %u vs %d", dsc.numTiles, n_tiles);
    AbstractDngDecompressor slices(r->img, dsc, compression, fix_ljpeg != 0,
                                   bps, /*predictor=*/1);
    slices.slices.reserve(n_tiles);
    for (int i = 0; i < n_tiles; ++i) {
      const Buffer b(tile_blobs[i],
                     implicit_cast<Buffer::size_type>(tile_bytes[i]));
      const ByteStream bs(DataBuffer(
          b, big_endian ? Endianness::big : Endianness::little));
      slices.slices.emplace_back(slices.dsc, i, bs);
    }
    slices.decompress();
  });
}

// Frames-parallel CPU baseline (the shape of rstest's `omp parallel for` over
// files, src/utilities/rstest/rstest.cpp:570): decode `n_frames` independent
// container blobs, each into its own image, with `threads` OpenMP threads.
// kind 0: LJpegDecoder full-image tile; kind 1: Cr2LJpegDecoder single slice.
int ref_ljpeg_frames_parallel(int n_frames, void* const* images,
                              const uint8_t* const* blobs,
                              const size_t* blob_bytes, int kind,
                              int threads) {
  int rc = RSX_OK;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < n_frames; ++i) {
    auto* r = static_cast<RefImage*>(images[i]);
    int st;
    if (kind == 0)
      st = ref_ljpeg_decode_container(images[i], blobs[i], blob_bytes[i], 0, 0,
                                      r->img->dim.x, r->img->dim.y,
                                      r->img->dim.x, r->img->dim.y, 0);
    else
      st = ref_cr2_decode_container(images[i], blobs[i], blob_bytes[i], 1, 0,
                                    r->img->dim.x);
    if (st != RSX_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc = st;
    }
  }
  return rc;
}

// The same fan-out at the decompressor level (what bench_ljpeg.py's plans are made
// of): `n_frames` independent entropy-coded scans, each with its own descriptor,
// decoded by LJpegDecompressor::decode (kind 0, descs = rsx_ljpeg_desc[]) or
// Cr2Decompressor::decompress (kind 1, descs = rsx_cr2_desc[]) on `threads` OpenMP
// threads -- one whole frame per thread, the decoders have no threading of their own.
int ref_scan_frames_parallel(int n_frames, void* const* images, const void* descs,
                             const uint8_t* const* ins, const size_t* in_bytes,
                             int kind, int threads) {
  int rc = RSX_OK;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < n_frames; ++i) {
    int st;
    if (kind == 0)
      st = ref_ljpeg_decompress(images[i],
                                static_cast<const rsx_ljpeg_desc*>(descs) + i, ins[i],
                                in_bytes[i], nullptr);
    else
      st = ref_cr2_decompress(images[i], static_cast<const rsx_cr2_desc*>(descs) + i,
                              ins[i], in_bytes[i], nullptr);
    if (st != RSX_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc = st;
    }
  }
  return rc;
}

int ref_unpack_frames_parallel(int n_frames, void* const* images,
                               const rsx_unpack_desc* d,
                               const uint8_t* const* ins, size_t in_bytes,
                               int threads) {
  int rc = RSX_OK;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < n_frames; ++i) {
    const int st = ref_unpack_u16(images[i], d, ins[i], in_bytes);
    if (st != RSX_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc = st;
    }
  }
  return rc;
}

} // extern "C"
