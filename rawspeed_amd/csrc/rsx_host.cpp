// rsx_host.cpp -- host-side logic of the core: descriptor validation (the
// reference constructors' checks), canonical Huffman table construction and
// output-geometry flattening.  No device code here.
#include "rsx_internal.h"

#include <cstring>
#include <map>
#include <mutex>

namespace rsx {

// ------------------------------------------------------------------------
// UncompressedDecompressor::UncompressedDecompressor
// (decompressors/UncompressedDecompressor.cpp:106-169), same order of checks.
// ------------------------------------------------------------------------
int validate_unpack(const rsx_unpack_desc& d, const rsx_image& img,
                    size_t in_bytes) {
  // input_.getStream(crop.dim.y, inputPitchBytes_) is the first member
  // initialiser: bounds / overflow -> IOException (io/ByteStream.h getStream).
  const uint64_t need = uint64_t(uint32_t(d.crop_h)) *
                        uint64_t(uint32_t(d.input_pitch_bytes));
  if (need > 0xFFFFFFFFull || need > in_bytes)
    return RSX_ERR_IO;
  if (d.crop_w <= 0 || d.crop_h <= 0) // "Empty tile." :112-113
    return RSX_ERR_INVALID_ARG;
  if (d.input_pitch_bytes < 1) // :115-116
    return RSX_ERR_INVALID_ARG;
  if (d.bit_order < RSX_ORDER_LSB || d.bit_order > RSX_ORDER_MSB32) // :118-127
    return RSX_ERR_INVALID_ARG;
  if (img.cpp < 1 || img.cpp > 3) // :135-136
    return RSX_ERR_INVALID_ARG;
  if (d.bits_per_pixel < 1 || d.bits_per_pixel > 16) // :138-140 (UINT16)
    return RSX_ERR_INVALID_ARG;
  const uint64_t bits =
      uint64_t(d.crop_w) * uint64_t(img.cpp) * uint64_t(d.bits_per_pixel);
  if (bits % 8 != 0) // :145-149
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.input_pitch_bytes) < bits / 8) // :155-156
    return RSX_ERR_INVALID_ARG;
  if (d.crop_x < 0 || d.crop_y < 0)
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.crop_y) > uint64_t(img.dim_y)) // :165-166
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.crop_x) + uint64_t(d.crop_w) > uint64_t(img.dim_x)) // :167-168
    return RSX_ERR_INVALID_ARG;
  // BitStreamerReplenisherBase ctor: "Bit stream size is smaller than
  // MaxProcessBytes" (bitstreams/BitStreamer.h:58-59); the 16-bit LSB
  // copyPixels path never builds a bit streamer (:255-265).
  const bool copy_path =
      d.bit_order == RSX_ORDER_LSB && d.bits_per_pixel == 16;
  if (!copy_path && need < 4)
    return RSX_ERR_IO;
  return RSX_OK;
}

// ------------------------------------------------------------------------
// HuffmanCode::setNCodesPerLength / setCodeValues (codes/HuffmanCode.h:99-166)
// + full-decode requirement (codes/AbstractPrefixCodeTranscoder.h:71-84).
// ------------------------------------------------------------------------
// ------------------------------------------------------------------------
// The same constructor for RawImageType::F32 images, then the dispatch of
// readUncompressedRaw (:212-245).
// ------------------------------------------------------------------------
int validate_unpack_f32(const rsx_unpack_desc& d, const rsx_image& img,
                        size_t in_bytes) {
  const uint64_t need = uint64_t(uint32_t(d.crop_h)) *
                        uint64_t(uint32_t(d.input_pitch_bytes));
  if (need > 0xFFFFFFFFull || need > in_bytes) // getStream, first initialiser
    return RSX_ERR_IO;
  if (d.crop_w <= 0 || d.crop_h <= 0) // :112-113
    return RSX_ERR_INVALID_ARG;
  if (d.input_pitch_bytes < 1) // :115-116
    return RSX_ERR_INVALID_ARG;
  if (d.bit_order < RSX_ORDER_LSB || d.bit_order > RSX_ORDER_MSB32) // :118-127
    return RSX_ERR_INVALID_ARG;
  if (img.cpp < 1 || img.cpp > 3) // :135-136
    return RSX_ERR_INVALID_ARG;
  if (d.bits_per_pixel < 1 || d.bits_per_pixel > 32) // :138-140 (F32 image)
    return RSX_ERR_INVALID_ARG;
  const uint64_t bits =
      uint64_t(d.crop_w) * uint64_t(img.cpp) * uint64_t(d.bits_per_pixel);
  if (bits % 8 != 0) // :145-149
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.input_pitch_bytes) < bits / 8) // :155-156
    return RSX_ERR_INVALID_ARG;
  if (d.crop_x < 0 || d.crop_y < 0)
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.crop_y) > uint64_t(img.dim_y)) // :165-166
    return RSX_ERR_INVALID_ARG;
  if (uint64_t(d.crop_x) + uint64_t(d.crop_w) > uint64_t(img.dim_x)) // :167-168
    return RSX_ERR_INVALID_ARG;
  // readUncompressedRaw :212-245
  if (d.bits_per_pixel == 32)
    return RSX_OK; // copyPixels, no bit streamer
  const bool byte_order = d.bit_order == RSX_ORDER_MSB || d.bit_order == RSX_ORDER_LSB;
  if (!byte_order || (d.bits_per_pixel != 16 && d.bits_per_pixel != 24))
    return RSX_ERR_INVALID_ARG; // "Unsupported floating-point input bitwidth/bit packing"
  if (need < 4) // the bit streamer refuses < 4 bytes (BitStreamer.h:58-59)
    return RSX_ERR_IO;
  return RSX_OK;
}

// ------------------------------------------------------------------------
// decode8BitRaw<true> / decode12BitRawWithControl<e> /
// decode12BitRawUnpackedLeftAligned<e>
// (decompressors/UncompressedDecompressor.cpp:270-378).  The constructor ran
// on the reference side already; these are the checks the methods add.
// ------------------------------------------------------------------------
int unpack_variant_bytes_per_line(const rsx_unpack_variant_desc& d, uint64_t* bpl) {
  const uint64_t w = uint64_t(uint32_t(d.w));
  switch (d.variant) {
  case RSX_UNPACK_8BIT_RAW: // sanityCheck(w, &h, 1) :273
  case RSX_UNPACK_8BIT_LOOKUP:
    *bpl = w;
    return RSX_OK;
  case RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED: // sanityCheck(w, &h, 2) :360
    *bpl = 2 * w;
    return RSX_OK;
  case RSX_UNPACK_12BIT_WITH_CONTROL:
    if ((12 * w) % 8 != 0) // bytesPerLine: ThrowIOE("Bad image width") :91-92
      return RSX_ERR_IO;
    *bpl = 12 * w / 8 + (w + 2) / 10; // :95-101
    return RSX_OK;
  default:
    return RSX_ERR_INVALID_ARG;
  }
}

int validate_unpack_variant(const rsx_unpack_variant_desc& d, const rsx_image& img,
                            size_t in_bytes) {
  if (d.variant < RSX_UNPACK_8BIT_RAW || d.variant > RSX_UNPACK_8BIT_LOOKUP)
    return RSX_ERR_INVALID_ARG;
  if (d.w <= 0 || d.h <= 0) // invariant(w > 0), invariant(*h > 0) :54, :78
    return RSX_ERR_INVALID_ARG;
  // out(row, col) addresses the uncropped u16 array (dim.x * cpp samples per
  // row); anything larger would be an out-of-bounds write in the reference
  if (img.cpp < 1 || img.dim_x <= 0 || img.dim_y <= 0 ||
      uint64_t(d.w) > uint64_t(img.dim_x) * uint64_t(img.cpp) || d.h > img.dim_y)
    return RSX_ERR_INVALID_ARG;
  uint64_t bpl = 0;
  if (int st = unpack_variant_bytes_per_line(d, &bpl))
    return st;
  // sanityCheck(h, bpl): fullRows = remain / bpl must reach h (:52-70), then
  // input.getData(bpl * h) (:277, :318, :363) -- both IOException
  if (bpl > 0x7FFFFFFFull || uint64_t(in_bytes) / bpl < uint64_t(d.h))
    return RSX_ERR_IO;
  return RSX_OK;
}

int validate_huff_table(const rsx_huff_table& t) {
  int max_len = 16;
  while (max_len > 0 && t.n_codes_per_length[max_len - 1] == 0)
    --max_len;
  if (max_len == 0)
    return RSX_ERR_INVALID_ARG; // "Codes-per-length table is empty"
  unsigned count = 0;
  for (int l = 1; l <= max_len; ++l)
    count += t.n_codes_per_length[l - 1];
  if (count > RSX_MAX_CODE_VALUES || count != t.n_code_values)
    return RSX_ERR_INVALID_ARG;
  unsigned max_codes = 2;
  for (int l = 1; l <= max_len; ++l) {
    const unsigned n = t.n_codes_per_length[l - 1];
    if (n > (1u << l) || n > max_codes)
      return RSX_ERR_INVALID_ARG; // "Corrupt Huffman"
    max_codes = (max_codes - n) * 2;
  }
  for (unsigned i = 0; i < count; ++i)
    if (t.code_values[i] > 16)
      return RSX_ERR_INVALID_ARG; // value is a difference length
  return RSX_OK;
}

// Canonical code -> device table.  Symbol semantics:
// AbstractPrefixCodeDecoder::processSymbol (AbstractPrefixCodeDecoder.h:43-66).
// NikonLASDecompressor (NikonDecompressor.cpp:79-377) takes any DHT-style table
// and has undefined behaviour for values it cannot decode; we accept what it can:
// a canonical code (no length overflowing its code space) whose values are 16
// (-32768), a plain SSSS 0..15, or len | shl << 4 with 0 < shl < len.
int validate_las_table(const rsx_huff_table& t) {
  unsigned total = 0;
  uint32_t code = 0;
  for (int l = 1; l <= 16; ++l) {
    const unsigned n = t.n_codes_per_length[l - 1];
    total += n;
    code += n;
    if (code > (1u << l))
      return RSX_ERR_INVALID_ARG;
    code <<= 1;
  }
  if (total == 0 || total > RSX_MAX_CODE_VALUES || total != t.n_code_values)
    return RSX_ERR_INVALID_ARG;
  for (unsigned i = 0; i < total; ++i) {
    const unsigned v = t.code_values[i];
    if (v != 16 && (v >> 4) != 0 && (v >> 4) >= (v & 15u))
      return RSX_ERR_INVALID_ARG; // getBits(len - shl <= 0) in the reference
  }
  return RSX_OK;
}

void build_device_table(const rsx_huff_table& t, DeviceHuffTable* out, bool las) {
  std::memset(out, 0, sizeof *out);
  out->las = las ? 1 : 0;
  int max_len = 16;
  while (max_len > 0 && t.n_codes_per_length[max_len - 1] == 0)
    --max_len;
  out->max_len = uint8_t(max_len);
  out->fix16 = t.fix_dng_bug16 ? 1 : 0;
  std::memcpy(out->values, t.code_values, t.n_code_values);
  for (int l = 0; l < 18; ++l)
    out->max_code[l] = 0xFFFFFFFFu;
  uint32_t code = 0;
  unsigned k = 0;
  for (int l = 1; l <= max_len; ++l) {
    const unsigned n = t.n_codes_per_length[l - 1];
    if (n) {
      out->val_offset[l] = uint16_t(code - k);
      out->max_code[l] = code + n - 1;
      for (unsigned i = 0; i < n; ++i, ++k, ++code) {
        const unsigned val = t.code_values[k];
        // LAS: the SSSS field holds len, the stream carries len - shl bits
        const unsigned ssss = (las && val != 16) ? (val & 15u) : val;
        unsigned total = l + (ssss == 16 ? (out->fix16 ? 16u : 0u)
                                         : (las ? ssss - (val >> 4) : ssss));
        if (l <= LUT_BITS) {
          const uint16_t e = uint16_t(l | (ssss << 5) | (total << 10));
          const uint32_t lo = code << (LUT_BITS - l);
          const uint32_t hi = lo | ((1u << (LUT_BITS - l)) - 1u);
          for (uint32_t c = lo; c <= hi; ++c)
            out->lut[c] = e;
        }
        if (code == 0 && l >= 1 && i == 0 && k == 0)
          out->zero_sym_bits = uint8_t(total); // all-zero code = first code
      }
    }
    code <<= 1;
  }
}

// ------------------------------------------------------------------------
// LJpegDecompressor::LJpegDecompressor (decompressors/LJpegDecompressor.cpp:52-152)
// ------------------------------------------------------------------------
static int validate_recipe(const rsx_huff_table* tables, int n_tables,
                           const uint8_t* table_index, int n_comp) {
  if (n_tables < 1 || n_tables > RSX_MAX_COMPONENTS)
    return RSX_ERR_INVALID_ARG;
  for (int i = 0; i < n_tables; ++i)
    if (int st = validate_huff_table(tables[i]))
      return st;
  for (int c = 0; c < n_comp; ++c)
    if (table_index[c] >= n_tables)
      return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

int validate_ljpeg(const rsx_ljpeg_desc& d, const rsx_image& img) {
  if (img.cpp < 1 || img.cpp > 3) // :61-68
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0) // :70-71
    return RSX_ERR_INVALID_ARG;
  if (d.tile_w <= 0 || d.tile_h <= 0) // :73-74
    return RSX_ERR_INVALID_ARG;
  if (d.tile_x < 0 || d.tile_y < 0)
    return RSX_ERR_INVALID_ARG;
  if (d.tile_x >= img.dim_x || d.tile_y >= img.dim_y) // :84-87
    return RSX_ERR_INVALID_ARG;
  if (d.tile_w > img.dim_x || d.tile_h > img.dim_y) // :89-92
    return RSX_ERR_INVALID_ARG;
  if (int64_t(d.tile_x) + d.tile_w > img.dim_x ||
      int64_t(d.tile_y) + d.tile_h > img.dim_y) // :94-97
    return RSX_ERR_INVALID_ARG;
  if (d.frame_w <= 0 || d.frame_h <= 0) // :99-100
    return RSX_ERR_INVALID_ARG;
  const int mw = d.mcu_w, mh = d.mcu_h;
  if (!((mh == 1 && mw >= 1 && mw <= 4) || (mw == 2 && mh == 2))) // :102-105
    return RSX_ERR_INVALID_ARG;
  if (d.n_comp != mw * mh) // :107-108
    return RSX_ERR_INVALID_ARG;
  if (d.rows_per_restart_interval < 1) // :115-116
    return RSX_ERR_INVALID_ARG;
  if (int64_t(mw) * d.frame_w > 0x7FFFFFFF ||
      int64_t(mh) * d.frame_h > 0x7FFFFFFF) // :118-122
    return RSX_ERR_INVALID_ARG;
  if (int64_t(img.cpp) * d.tile_w > 0x7FFFFFFF) // "Img frame is too big" :124-126
    return RSX_ERR_INVALID_ARG;
  if (d.tile_w < mw || d.tile_h < mh) // :128-129
    return RSX_ERR_INVALID_ARG;
  if (d.tile_h % mh != 0) // :131-132
    return RSX_ERR_INVALID_ARG;
  const int64_t req_w = int64_t(img.cpp) * d.tile_w;
  const int64_t mcus_to_consume = (req_w + mw - 1) / mw;
  if (d.frame_w < mcus_to_consume || int64_t(mh) * d.frame_h < d.tile_h ||
      int64_t(mw) * d.frame_w < req_w) // :137-146
    return RSX_ERR_INVALID_ARG;
  return validate_recipe(d.tables, d.n_tables, d.table_index, d.n_comp);
}

// ------------------------------------------------------------------------
// Cr2Decompressor geometry (Cr2DecompressorImpl.h:76-244, 279-363)
// ------------------------------------------------------------------------
namespace {

struct Rect {
  int x, y, w, h;
};

struct Cr2Geom {
  int N, xsf, ysf, sub, slice_col_step, px_per_group, group_size;
  int dim_x, dim_y, frame_x, frame_y, n_slices, slice_w, last_w;
};

// evaluateConsecutiveTiles :60-72
int consecutive(const Rect& a, const Rect& b) {
  if (a.x == b.x && a.y + a.h == b.y && a.w == b.w)
    return 1; // ContinuesColumn
  if (b.y == 0 && b.x == a.x + a.w)
    return 2; // BeginsNewColumn
  return 0;
}

// Cr2OutputTileIterator :104-154, materialised.
std::vector<Rect> all_output_tiles(const Cr2Geom& g) {
  std::vector<Rect> tiles;
  int ox = 0, oy = 0, slice_row = 0, id = 0;
  while (id < g.n_slices) {
    Rect t{ox, oy, id + 1 == g.n_slices ? g.last_w : g.slice_w, 0};
    const int out_rem = g.dim_y - oy, tile_rem = g.frame_y - slice_row;
    t.h = out_rem < tile_rem ? out_rem : tile_rem;
    tiles.push_back(t);
    slice_row += t.h;
    oy += t.h;
    if (slice_row == g.frame_y) {
      ++id;
      slice_row = 0;
    }
    if (oy == g.dim_y) {
      oy = 0;
      // a column that starts at or past the right edge holds nothing of the image: the
      // reference's walk stops at the first tile that lies outside it
      // (Cr2DecompressorImpl.h:404-407), and ox must not run away over many wide slices
      if (int64_t(ox) + t.w >= int64_t(g.dim_x))
        break;
      ox += t.w;
    }
    if (t.h <= 0 && tiles.size() > 100000)
      break; // defensive: cannot happen for validated geometry
  }
  return tiles;
}

int cr2_geom(const rsx_cr2_desc& d, const rsx_image& img, Cr2Geom* g) {
  if (d.num_slices < 1) // Cr2SliceWidths ctor, Cr2Decompressor.h:66-67
    return RSX_ERR_INVALID_ARG;
  if (img.cpp != 1) // :290-291
    return RSX_ERR_INVALID_ARG;
  const int N = d.n_comp, X = d.x_s_f, Y = d.y_s_f;
  if (!((N == 3 && X == 2 && Y == 2) || (N == 3 && X == 2 && Y == 1) ||
        (N == 2 && X == 1 && Y == 1) || (N == 4 && X == 1 && Y == 1))) // :293-298
    return RSX_ERR_INVALID_ARG;
  g->N = N;
  g->xsf = X;
  g->ysf = Y;
  g->sub = (X != 1 || Y != 1);
  g->slice_col_step = N * X;
  g->px_per_group = X * Y;
  g->group_size = !g->sub ? N : 2 + g->px_per_group; // Dsc :250-275
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_x % g->group_size != 0) // :300-302
    return RSX_ERR_INVALID_ARG;
  g->dim_x = img.dim_x / g->group_size;
  g->dim_y = img.dim_y;
  if (d.frame_w <= 0 || d.frame_h <= 0 || d.frame_w % X != 0 ||
      d.frame_h % Y != 0) // :305-308
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x > 19440 || img.dim_y > 5920) // :313-316
    return RSX_ERR_INVALID_ARG;
  for (int i = 0; i < d.num_slices; ++i) { // :318-322
    const int w = i + 1 == d.num_slices ? d.last_slice_width : d.slice_width;
    if (w <= 0)
      return RSX_ERR_INVALID_ARG;
  }
  if (g->sub == (img.is_cfa != 0)) // :324-325
    return RSX_ERR_INVALID_ARG;
  if (int st = validate_recipe(d.tables, d.n_tables, d.table_index, N)) // :327-333
    return st;
  if (d.slice_width % g->slice_col_step != 0 ||
      d.last_slice_width % g->slice_col_step != 0) // :335-341
    return RSX_ERR_INVALID_ARG;
  g->frame_x = d.frame_w / X;
  g->frame_y = d.frame_h / Y;
  g->n_slices = d.num_slices;
  g->slice_w = d.slice_width / g->slice_col_step;
  g->last_w = d.last_slice_width / g->slice_col_step;
  if (int64_t(g->frame_x) * g->frame_y < int64_t(g->dim_x) * g->dim_y) // :343-344
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

// ctor tiling checks :346-362; returns the tiles that contribute
int cr2_output_tiles(const Cr2Geom& g, std::vector<Rect>* out) {
  const std::vector<Rect> all = all_output_tiles(g);
  bool have_last = false;
  Rect last{};
  size_t n_used = 0;
  for (size_t i = 0; i < all.size(); ++i) {
    const Rect& t = all[i];
    if (have_last && consecutive(last, t) == 0)
      return RSX_ERR_INVALID_ARG; // "Invalid tiling"
    if (t.x + t.w <= g.dim_x && t.y + t.h <= g.dim_y) {
      last = t;
      have_last = true;
      n_used = i + 1;
      continue;
    }
    if (t.x < g.dim_x && t.y < g.dim_y)
      return RSX_ERR_INVALID_ARG; // "Output tile partially outside of image"
    break;
  }
  if (!have_last)
    return RSX_ERR_INVALID_ARG; // "No tiles are provided"
  if (last.x + last.w != g.dim_x || last.y + last.h != g.dim_y)
    return RSX_ERR_INVALID_ARG; // "Tiles do not cover the entire image area."
  // getOutputTiles :224-231: up to and including the first tile whose
  // bottom-right corner is the image's.
  out->clear();
  for (size_t i = 0; i < n_used; ++i) {
    out->push_back(all[i]);
    if (all[i].x + all[i].w == g.dim_x && all[i].y + all[i].h == g.dim_y)
      break;
  }
  return RSX_OK;
}

} // namespace

int validate_cr2(const rsx_cr2_desc& d, const rsx_image& img) {
  Cr2Geom g;
  if (int st = cr2_geom(d, img, &g))
    return st;
  std::vector<Rect> tiles;
  return cr2_output_tiles(g, &tiles);
}

int build_ljpeg_stream(const rsx_ljpeg_desc& d, const rsx_image& img,
                       StreamGeom* s) {
  if (int st = validate_ljpeg(d, img))
    return st;
  std::memset(s, 0, sizeof *s);
  s->kind = 0;
  s->n_comp = uint32_t(d.n_comp);
  s->period = uint32_t(d.n_comp);
  for (int c = 0; c < d.n_comp; ++c) {
    s->comp_of_phase[c] = d.table_index[c];
    s->pred_of_phase[c] = uint8_t(c);
    s->init_pred[c] = d.init_pred[c];
    s->seed_pos[c] = uint8_t(c);
  }
  s->row_samples = uint32_t(d.frame_w) * uint32_t(d.n_comp);
  s->rows = uint32_t(d.tile_h / d.mcu_h); // rows below the tile are never decoded (:312-315)
  s->mcu_w = uint32_t(d.mcu_w);
  s->mcu_h = uint32_t(d.mcu_h);
  s->out_x = uint32_t(img.cpp) * uint32_t(d.tile_x);
  s->out_y = uint32_t(d.tile_y);
  s->keep_samples = uint32_t(img.cpp) * uint32_t(d.tile_w);
  s->img_pitch_bytes = img.pitch_bytes;
  return RSX_OK;
}

int build_cr2_stream(const rsx_cr2_desc& d, const rsx_image& img,
                     StreamGeom* s) {
  Cr2Geom g;
  if (int st = cr2_geom(d, img, &g))
    return st;
  std::vector<Rect> tiles;
  if (int st = cr2_output_tiles(g, &tiles))
    return st;
  std::memset(s, 0, sizeof *s);
  s->kind = 1;
  s->n_comp = uint32_t(g.N);
  // symbol p of a group belongs to component (p < pixelsPerGroup ? 0 :
  // p - pixelsPerGroup + 1) (:457-458); for <N,1,1> that is p itself
  s->period = uint32_t(g.group_size);
  for (int p = 0; p < g.group_size; ++p) {
    const int c = !g.sub ? p : (p < g.px_per_group ? 0 : p - g.px_per_group + 1);
    s->comp_of_phase[p] = d.table_index[c];
    s->pred_of_phase[p] = uint8_t(c);
  }
  for (int c = 0; c < g.N; ++c) {
    s->init_pred[c] = d.init_pred[c];
    // the row predictor comes from predNext(c == 0 ? 0 : groupSize - (N - c)) :443-444
    s->seed_pos[c] = uint8_t(c == 0 ? 0 : g.group_size - (g.N - c));
  }
  s->row_samples = uint32_t(g.frame_x) * uint32_t(g.group_size);
  // Only the groups that land in the image are decoded: dim.area() groups
  // (:431-465 walks output strips, not the frame).
  const uint64_t total_groups = uint64_t(g.dim_x) * uint64_t(g.dim_y);
  s->rows = uint32_t((total_groups + g.frame_x - 1) / g.frame_x);
  s->mcu_w = uint32_t(g.group_size);
  s->mcu_h = 1;
  s->img_pitch_bytes = img.pitch_bytes;
  // coalesce vertically adjacent tiles into strips (:156-205)
  std::vector<Rect> strips;
  for (const Rect& t : tiles) {
    if (!strips.empty() && consecutive(strips.back(), t) == 1 &&
        strips.back().y + strips.back().h == t.y)
      strips.back().h += t.h;
    else
      strips.push_back(t);
  }
  if (strips.size() > size_t(MAX_CR2_STRIPS))
    return RSX_ERR_UNSUPPORTED;
  s->n_strips = uint32_t(strips.size());
  uint64_t first = 0;
  for (size_t k = 0; k < strips.size(); ++k) {
    s->strip_x0[k] = uint32_t(strips[k].x) * uint32_t(g.group_size);
    s->strip_w[k] = uint32_t(strips[k].w) * uint32_t(g.group_size);
    s->strip_y0[k] = uint32_t(strips[k].y);
    s->strip_h[k] = uint32_t(strips[k].h);
    s->strip_first_sample[k] = first;
    first += uint64_t(s->strip_w[k]) * uint64_t(s->strip_h[k]);
  }
  s->strip_first_sample[strips.size()] = first;
  return RSX_OK;
}

// ------------------------------------------------------------------------
// NikonDecompressor::NikonDecompressor (decompressors/NikonDecompressor.cpp:473-513)
// -- the part of it that is about the image and the decode parameters; the
// metadata parsing itself (v0/v1, curve, split) stays in the reference.
// ------------------------------------------------------------------------
int validate_nikon(const rsx_nikon_desc& d, const rsx_image& img) {
  if (img.cpp != 1) // :476-478
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_x % 2 != 0 || img.dim_x > 8288 ||
      img.dim_y > 5520) // :480-483
    return RSX_ERR_INVALID_ARG;
  if (d.bits_ps != 12 && d.bits_ps != 14) // :485-491
    return RSX_ERR_INVALID_ARG;
  // "If the 'split' happens outside of the image, it does not actually
  // happen" (:511-512): the caller passes the clamped value
  if (d.split < 0 || d.split >= img.dim_y)
    return RSX_ERR_INVALID_ARG;
  for (int i = 0; i < 4; ++i) // metadata.getU16() :506-509
    if ((&d.p_up[0][0])[i] < 0 || (&d.p_up[0][0])[i] > 65535)
      return RSX_ERR_INVALID_ARG;
  if (!d.uncorrected_raw_values &&
      (d.curve == nullptr || d.curve_size < 1 || d.curve_size > 65536))
    return RSX_ERR_INVALID_ARG; // TableLookUp::setTable (TableLookUp.cpp:50-57)
  if (int st = validate_huff_table(d.tables[0])) // createPrefixCodeDecoder :455-470
    return st;
  if (d.tables[0].fix_dng_bug16) // ht.setup(true, false)
    return RSX_ERR_INVALID_ARG;
  if (d.split != 0)
    if (int st = validate_las_table(d.tables[1]))
      return st;
  return RSX_OK;
}

// PentaxDecompressor::PentaxDecompressor (decompressors/PentaxDecompressor.cpp:55-67)
int validate_pentax(const rsx_pentax_desc& d, const rsx_image& img) {
  if (img.cpp != 1) // :58-60
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_x % 2 != 0 || img.dim_x > 8384 ||
      img.dim_y > 6208) // :62-66
    return RSX_ERR_INVALID_ARG;
  if (int st = validate_huff_table(d.table)) // SetupPrefixCodeDecoder :139-150
    return st;
  if (d.table.fix_dng_bug16) // ht.setup(true, false)
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

// SamsungV1Decompressor::SamsungV1Decompressor (SamsungV1Decompressor.cpp:45-61)
// and the table its decompress() builds (:88-117)
int validate_samsung_v1(const rsx_samsung_v1_desc& d, const rsx_image& img) {
  if (img.cpp != 1) // :48-50
    return RSX_ERR_INVALID_ARG;
  if (d.bits != 12) // :53-54
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_x % 32 != 0 || img.dim_y % 2 != 0 ||
      img.dim_x > 5664 || img.dim_y > 3714) // :59-61
    return RSX_ERR_INVALID_ARG;
  // the pairs must tile the 1024-entry table exactly, as the reference's do
  if (d.n_entries < 1 || d.n_entries > RSX_SAMSUNG_V1_MAX_ENTRIES)
    return RSX_ERR_INVALID_ARG;
  uint32_t filled = 0;
  for (int i = 0; i < d.n_entries; ++i) {
    if (d.enc_len[i] < 1 || d.enc_len[i] > 10 || d.diff_len[i] > 13)
      return RSX_ERR_INVALID_ARG; // fill(23) covers at most 10 + 13 bits (:66)
    filled += 1024u >> d.enc_len[i];
  }
  return filled == 1024 ? RSX_OK : RSX_ERR_INVALID_ARG;
}

void build_device_table_explicit(const uint8_t* enc_len, const uint8_t* diff_len, int n,
                                 DeviceHuffTable* out, int bits) {
  std::memset(out, 0, sizeof *out);
  for (int l = 0; l < 18; ++l)
    out->max_code[l] = 0xFFFFFFFFu;
  out->max_len = uint8_t(bits);
  const uint32_t size = 1u << bits, rep = 1u << (LUT_BITS - bits);
  uint32_t pos = 0; // index into the 2^bits-entry table of the reference
  for (int i = 0; i < n; ++i) {
    const uint32_t l = enc_len[i], ssss = diff_len[i];
    const bool invalid = diff_len[i] == 0xFF;
    const uint16_t e = invalid ? uint16_t(0) : uint16_t(l | (ssss << 5) | ((l + ssss) << 10));
    const uint32_t cnt = size >> l;
    // every index of the reference's table covers `rep` LUT slots
    for (uint32_t c = pos * rep; c < (pos + cnt) * rep; ++c)
      out->lut[c] = e;
    if (pos == 0)
      out->zero_sym_bits = invalid ? uint8_t(0) : uint8_t(l + ssss);
    pos += cnt;
  }
}

// SonyArw1Decompressor::SonyArw1Decompressor (decompressors/SonyArw1Decompressor.cpp:39-51)
int validate_sony_arw1(const rsx_image& img) {
  if (img.cpp != 1) // :41-43
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_y % 2 != 0 || img.dim_x > 4600 ||
      img.dim_y > 3072) // :48-49
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

// Cr2sRawInterpolator::interpolate (interpolators/Cr2sRawInterpolator.cpp:510-542) and
// the shapes Cr2Decoder::sRawInterpolate sets up (Cr2Decoder.cpp:589-601)
int validate_sraw(const rsx_sraw_desc& d, const rsx_image& in, const rsx_image& out) {
  if (d.version < 0 || d.version > 2) // invariant :511
    return RSX_ERR_INVALID_ARG;
  if (d.subsampling_y != 1 && d.subsampling_y != 2) // "Unknown subsampling" :540-541
    return RSX_ERR_INVALID_ARG;
  if (d.subsampling_y == 2 && d.version == 0) // "no known sraws with version 0" :529-538
    return RSX_ERR_INVALID_ARG;
  const int gs = 2 + 2 * d.subsampling_y;
  if (in.cpp != 1 || out.cpp != 3)
    return RSX_ERR_INVALID_ARG;
  if (in.dim_x <= 0 || in.dim_y <= 0 || in.dim_x % gs != 0) // :104, :201
    return RSX_ERR_INVALID_ARG;
  const int mcus = in.dim_x / gs;
  if (mcus <= 1) // invariant(numMCUs > 1) :106, :203
    return RSX_ERR_INVALID_ARG;
  // Cr2Decoder.cpp:589-596: the output is 2 pixels per group wide, y * rows high
  if (out.dim_x != 2 * mcus || out.dim_y != d.subsampling_y * in.dim_y)
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

// HasselbladDecompressor::HasselbladDecompressor (HasselbladDecompressor.cpp:37-57) and
// ht.verifyCodeValuesAsDiffLengths() (:80)
int validate_hasselblad(const rsx_hasselblad_desc& d, const rsx_image& img) {
  if (img.cpp != 1) // :44-45
    return RSX_ERR_INVALID_ARG;
  if (img.dim_x <= 0 || img.dim_y <= 0 || img.dim_x % 2 != 0 || img.dim_x > 12000 ||
      img.dim_y > 8842) // :48-52
    return RSX_ERR_INVALID_ARG;
  if (int st = validate_huff_table(d.table))
    return st;
  if (d.table.fix_dng_bug16)
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

// TableLookUp::setTable, dither branch (common/TableLookUp.cpp:66-84).  Only
// the first 32768 entries can be addressed: the index is clampBits(pred, 15).
void build_dither_table(const uint16_t* curve, int n, std::vector<uint32_t>* out) {
  out->assign(32768, 0u);
  for (int i = 0; i < 32768; ++i) {
    if (i < n) {
      const int center = curve[i];
      int lower = i > 0 ? curve[i - 1] : center;
      int upper = i < n - 1 ? curve[i + 1] : center;
      lower = lower < center ? lower : center;
      upper = upper > center ? upper : center;
      const int delta = upper - lower;
      int base = center - ((upper - lower + 2) / 4);
      base = base < 0 ? 0 : (base > 65535 ? 65535 : base); // clampBits(.., 16)
      (*out)[i] = uint32_t(base) | (uint32_t(delta) << 16);
    } else {
      (*out)[i] = curve[n - 1];
    }
  }
}

// Device allocations are recycled: a host-pointer call builds and drops a plan
// (some 20 buffers) per image, and hipMalloc / hipFree (which synchronises the
// device) would otherwise cost more than the kernels.  Freed blocks are kept per
// device, keyed by size, up to a cap; a request takes the smallest cached block
// that is large enough and at most twice what it asked for.  Callers release a
// buffer only after the stream that used it has been synchronised.
namespace {

struct BlockCache {
  static constexpr int kMaxDevices = 64;
  // (freed device blocks kept for reuse by the next plan: a batch pipeline recycles a
  // few hundred MB per image; anything beyond this goes back to the driver at once)
  static constexpr size_t kMaxCachedBytes = size_t(2) << 30;
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks[kMaxDevices];
  size_t cached_bytes = 0;
};

BlockCache& block_cache() {
  static BlockCache* c = new BlockCache; // never destroyed: the HIP runtime may
  return *c;                             // already be gone at static-destruction time
}

int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= BlockCache::kMaxDevices)
    return 0;
  return d;
}

} // namespace

int DeviceBuffer::ensure(size_t n) {
  if (n <= bytes)
    return RSX_OK;
  release();
  // size classes: 64 KiB steps below 1 MiB, 1 MiB steps above
  const size_t step = n < (size_t(1) << 20) ? (size_t(1) << 16) : (size_t(1) << 20);
  const size_t want = (n + step - 1) / step * step;
  {
    BlockCache& c = block_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    auto& m = c.free_blocks[current_device()];
    auto it = m.lower_bound(want);
    if (it != m.end() && it->first <= 2 * want) {
      ptr = it->second;
      bytes = it->first;
      c.cached_bytes -= it->first;
      m.erase(it);
      return RSX_OK;
    }
  }
  if (hipMalloc(&ptr, want) != hipSuccess) {
    // give the cache back to the driver and retry once
    BlockCache& c = block_cache();
    {
      std::lock_guard<std::mutex> lock(c.mu);
      auto& m = c.free_blocks[current_device()];
      for (auto& kv : m) {
        (void)hipFree(kv.second);
        c.cached_bytes -= kv.first;
      }
      m.clear();
    }
    if (hipMalloc(&ptr, want) != hipSuccess) {
      ptr = nullptr;
      bytes = 0;
      return RSX_ERR_NOMEM;
    }
  }
  bytes = want;
  return RSX_OK;
}

void DeviceBuffer::release() {
  if (ptr) {
    BlockCache& c = block_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.cached_bytes + bytes <= BlockCache::kMaxCachedBytes) {
      c.free_blocks[current_device()].emplace(bytes, ptr);
      c.cached_bytes += bytes;
    } else {
      (void)hipFree(ptr);
    }
  }
  ptr = nullptr;
  bytes = 0;
}

} // namespace rsx
