/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * rsx_oracle.c: a plain-C, single-threaded CPU restatement of the reference
 * algorithm (darktable-org/rawspeed @ 2024-10-16) for the hot path this
 * repository accelerates.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.  The product
 * (rawspeed_amd/csrc) never links, imports or calls anything in this file.
 *
 * Parity pin: this restatement is validated (tests/test_oracle_*.py) against
 *   (a) the reference's own known-answer vectors for the layers under the
 *       path (bit readers: test/librawspeed/bitstreams/BitStreamer*Test.cpp,
 *       Huffman: test/librawspeed/codes/HuffmanTableTest.cpp), and
 *   (b) the unmodified reference compiled into oracle/_ref/ (oracle/Makefile,
 *       oracle/ref_shim.cpp) on seeded synthetic inputs, plus golden output
 *       hashes generated from that build and committed under tests/golden/.
 *
 * Each function cites the reference file:line (relative to
 * src/librawspeed/) it follows.  The code is written from the behaviour of
 * those lines, sequentially and without any of the reference's data
 * structures (no LUT decoder, no iterators, no templates).
 */
#include "rsx.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* Bit streamer: bitstreams/BitStreamer.h + BitStream.h + per-order traits.   */
/* ======================================================================== */

typedef struct bitreader {
  const uint8_t* data;
  int64_t size;   /* input.size() */
  int64_t pos;    /* replenisher pos (BitStreamer.h:48) */
  uint64_t cache; /* BitStreamCacheBase::cache (BitStream.h:42) */
  int fill;       /* BitStreamCacheBase::fillLevel */
  int order;      /* rsx_bit_order */
  int64_t eos;    /* BitStreamerJPEG::endOfStreamPos, -1 = unknown */
  int err;        /* sticky rsx_status */
} bitreader;

static int max_process_bytes(int order) {
  /* BitStreamerTraits<>::MaxProcessBytes: 4 everywhere, 8 for JPEG
   * (BitStreamerJPEG.h:74-79). */
  return order == RSX_ORDER_JPEG ? 8 : 4;
}

/* BitStreamerReplenisherBase ctor (BitStreamer.h:56-60). */
static void br_init(bitreader* b, const uint8_t* data, int64_t size,
                    int order) {
  b->data = data;
  b->size = size;
  b->pos = 0;
  b->cache = 0;
  b->fill = 0;
  b->order = order;
  b->eos = -1;
  b->err = size < max_process_bytes(order) ? RSX_ERR_IO : RSX_OK;
}

/* BitStreamerForwardSequentialReplenisher::getInput (BitStreamer.h:100-132)
 * with the zero-padded tail load of adt/VariableLengthLoad.h:148-175. */
static void br_get_input(bitreader* b, uint8_t tmp[8]) {
  const int n = max_process_bytes(b->order);
  memset(tmp, 0, 8);
  if (b->pos + n <= b->size) {
    memcpy(tmp, b->data + b->pos, (size_t)n);
    return;
  }
  if (b->pos > b->size + 2 * n) {
    b->err = RSX_ERR_INPUT_OVERFLOW; /* BitStreamer.h:125-127 */
    return;
  }
  int64_t from = b->pos < b->size ? b->pos : b->size;
  int64_t to = from + n < b->size ? from + n : b->size;
  if (to > from)
    memcpy(tmp, b->data + from, (size_t)(to - from));
}

/* BitStreamCacheLeftInRightOut::push / RightInLeftOut::push
 * (BitStream.h:59-67, :92-117). */
static void br_push(bitreader* b, uint64_t bits, int count) {
  if (b->order == RSX_ORDER_LSB) {
    b->cache |= bits << b->fill;
  } else if (count != 0) {
    b->cache |= bits << (64 - b->fill - count);
  }
  b->fill += count;
}

static uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
static uint32_t le32(const uint8_t* p) {
  return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 |
         (uint32_t)p[3] << 24;
}
static uint32_t be32(const uint8_t* p) {
  return (uint32_t)p[3] | (uint32_t)p[2] << 8 | (uint32_t)p[1] << 16 |
         (uint32_t)p[0] << 24;
}

/* BitStreamer::fillCache (BitStreamer.h:155-182) with the chunk type and
 * endianness of BitStreamLSB.h / BitStreamMSB.h / BitStreamMSB16.h /
 * BitStreamMSB32.h :31-43, and BitStreamerJPEG::fillCache
 * (BitStreamerJPEG.h:106-183).  Returns bytes consumed. */
static int64_t br_fill_cache(bitreader* b, const uint8_t in[8]) {
  switch (b->order) {
  case RSX_ORDER_LSB:
    br_push(b, le32(in), 32);
    return 4;
  case RSX_ORDER_MSB:
    br_push(b, be32(in), 32);
    return 4;
  case RSX_ORDER_MSB16:
    br_push(b, le16(in), 16);
    br_push(b, le16(in + 2), 16);
    return 4;
  case RSX_ORDER_MSB32:
    br_push(b, le32(in), 32);
    return 4;
  default:
    break;
  }
  /* JPEG */
  if (in[0] != 0xFF && in[1] != 0xFF && in[2] != 0xFF && in[3] != 0xFF) {
    br_push(b, be32(in), 32); /* :121-129 */
    return 4;
  }
  int64_t p = 0;
  for (int i = 0; i < 4; ++i) {
    const uint8_t c0 = in[p];
    br_push(b, c0, 8);
    if (c0 != 0xFF) {
      p += 1;
      continue;
    }
    if (in[p + 1] == 0x00) { /* FF 00: stuffed data byte FF (:145-151) */
      p += 2;
      continue;
    }
    /* FF xx: end of stream (:153-179) */
    b->eos = b->pos + p;
    b->fill -= 8;
    b->cache &= ~((~0ULL) >> b->fill);
    b->fill = 64;
    p = (b->size - b->pos) + (4 - i);
    break;
  }
  return p;
}

/* BitStreamer::fill (BitStreamer.h:216-229). */
static void br_fill(bitreader* b, int nbits) {
  if (b->fill >= nbits)
    return;
  uint8_t tmp[8];
  br_get_input(b, tmp);
  if (b->err)
    return;
  b->pos += br_fill_cache(b, tmp);
}

/* peekBitsNoFill / skipBitsNoFill (BitStreamer.h:253-268, BitStream.h:69-89,
 * :119-140). */
static uint32_t br_peek_nofill(const bitreader* b, int n) {
  if (b->order == RSX_ORDER_LSB)
    return (uint32_t)b->cache & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
  return (uint32_t)(b->cache >> (64 - n));
}
static void br_skip_nofill(bitreader* b, int n) {
  if (n == 0)
    return;
  if (b->order == RSX_ORDER_LSB)
    b->cache = n >= 64 ? 0 : b->cache >> n;
  else
    b->cache = n >= 64 ? 0 : b->cache << n;
  b->fill -= n;
}
static uint32_t br_get_nofill(bitreader* b, int n) {
  uint32_t v = br_peek_nofill(b, n);
  br_skip_nofill(b, n);
  return v;
}
/* getBits (BitStreamer.h:294-301). */
static uint32_t br_get(bitreader* b, int n) {
  br_fill(b, n);
  return br_get_nofill(b, n);
}
/* skipManyBits / skipBytes (BitStreamer.h:305-325). */
static void br_skip_bytes(bitreader* b, int64_t nbytes) {
  int64_t rem = 8 * nbytes;
  for (; rem >= 32 && !b->err; rem -= 32) {
    br_fill(b, 32);
    br_skip_nofill(b, 32);
  }
  if (rem > 0 && !b->err) {
    br_fill(b, (int)rem);
    br_skip_nofill(b, (int)rem);
  }
}
/* getStreamPosition: BitStreamerJPEG.h:185-189 (marker position once seen,
 * else the input position). */
static int64_t br_jpeg_stream_position(const bitreader* b) {
  return b->eos >= 0 ? b->eos : b->pos;
}

/* ---- exported probes for the known-answer tests of the bit readers ------ */

/* Reads `n` values of `lens[i]` bits each with getBits() from a fresh
 * streamer of the given order (test/librawspeed/bitstreams/BitStreamerTest.h
 * :171-243 use exactly this access pattern).  peek_only[i] != 0 => peekBits
 * then skipBits is used instead (same result). */
int oracle_bitreader_get(int order, const uint8_t* data, size_t size, int n,
                         const int32_t* lens, uint32_t* out) {
  bitreader b;
  br_init(&b, data, (int64_t)size, order);
  if (b.err)
    return b.err;
  for (int i = 0; i < n; ++i) {
    out[i] = br_get(&b, lens[i]);
    if (b.err)
      return b.err;
  }
  return RSX_OK;
}

/* peekBits(len) for len = 1..n without consuming (IncreasingPeekLengthTest,
 * BitStreamerTest.h:97-110). */
int oracle_bitreader_peek_increasing(int order, const uint8_t* data, size_t size,
                                     int n, uint32_t* out) {
  bitreader b;
  br_init(&b, data, (int64_t)size, order);
  if (b.err)
    return b.err;
  for (int len = 1; len <= n; ++len) {
    br_fill(&b, len);
    if (b.err)
      return b.err;
    out[len - 1] = br_peek_nofill(&b, len);
  }
  return RSX_OK;
}

/* ======================================================================== */
/* UncompressedDecompressor                                                   */
/* ======================================================================== */

/* Constructor checks, in the reference's order
 * (UncompressedDecompressor.cpp:106-169; ByteStream::getStream bounds check
 * happens first because `input` is the first member initialised). */
int oracle_unpack_validate(const rsx_unpack_desc* d, const rsx_image* img,
                           size_t in_bytes) {
  const uint64_t need =
      (uint64_t)(uint32_t)d->crop_h * (uint64_t)(uint32_t)d->input_pitch_bytes;
  if (need > 0xFFFFFFFFull || need > in_bytes)
    return RSX_ERR_IO; /* io/ByteStream.h getStream -> check */
  if (d->crop_w <= 0 || d->crop_h <= 0)
    return RSX_ERR_INVALID_ARG; /* "Empty tile." :112-113 */
  if (d->input_pitch_bytes < 1)
    return RSX_ERR_INVALID_ARG; /* :115-116 */
  if (d->bit_order < RSX_ORDER_LSB || d->bit_order > RSX_ORDER_MSB32)
    return RSX_ERR_INVALID_ARG; /* :118-127 */
  if (img->cpp < 1 || img->cpp > 3)
    return RSX_ERR_INVALID_ARG; /* :135-136 */
  if (d->bits_per_pixel < 1 || d->bits_per_pixel > 16)
    return RSX_ERR_INVALID_ARG; /* :138-140 (UINT16 image) */
  const uint64_t bits =
      (uint64_t)d->crop_w * (uint64_t)img->cpp * (uint64_t)d->bits_per_pixel;
  if (bits % 8 != 0)
    return RSX_ERR_INVALID_ARG; /* :145-149 */
  if ((uint64_t)d->input_pitch_bytes < bits / 8)
    return RSX_ERR_INVALID_ARG; /* :155-156 */
  if (d->crop_y < 0 || d->crop_x < 0)
    return RSX_ERR_INVALID_ARG;
  if ((uint64_t)d->crop_y > (uint64_t)img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :165-166 */
  if ((uint64_t)d->crop_x + (uint64_t)d->crop_w > (uint64_t)img->dim_x)
    return RSX_ERR_INVALID_ARG; /* :167-168 */
  return RSX_OK;
}

/* readUncompressedRaw for UINT16 images (UncompressedDecompressor.cpp:202-268)
 * -> decodePackedInt<Pump> (:188-200) or the 16-bit-LSB copyPixels path
 * (:255-265, common/Common.h:79-93). */
int oracle_unpack_u16(const rsx_unpack_desc* d, const uint8_t* in,
                      size_t in_bytes, const rsx_image* img) {
  int st = oracle_unpack_validate(d, img, in_bytes);
  if (st)
    return st;
  uint8_t* out = (uint8_t*)img->data;
  const int64_t cols = (int64_t)d->crop_w * img->cpp;
  const int64_t stream_bytes = (int64_t)d->crop_h * d->input_pitch_bytes;
  const int64_t skip =
      d->input_pitch_bytes - cols * d->bits_per_pixel / 8; /* :162-163 */
  int64_t y = d->crop_y;
  int64_t h = (int64_t)d->crop_h + d->crop_y; /* :209-210 */
  if (h > img->dim_y)
    h = img->dim_y;

  if (d->bit_order == RSX_ORDER_LSB && d->bits_per_pixel == 16) {
    /* strided memcpy, honours offset.x (:257-264) */
    for (int64_t r = y; r < h; ++r)
      memcpy(out + r * img->pitch_bytes + (size_t)d->crop_x * img->cpp * 2,
             in + (r - y) * d->input_pitch_bytes, (size_t)cols * 2);
    return RSX_OK;
  }

  bitreader b;
  br_init(&b, in, stream_bytes, d->bit_order);
  if (b.err)
    return b.err;
  for (int64_t row = y; row < h; ++row) {
    uint16_t* orow = (uint16_t*)(out + row * img->pitch_bytes);
    /* NOTE: column index ignores offset.x (:196), replicated on purpose. */
    for (int64_t x = 0; x < cols; ++x)
      orow[x] = (uint16_t)br_get(&b, d->bits_per_pixel);
    br_skip_bytes(&b, skip);
    if (b.err)
      return b.err;
  }
  return RSX_OK;
}

/* ---- RawImageType::F32 images ------------------------------------------------ */

/* extendBinaryFloatingPoint<Narrow, Binary32> (common/FloatingPoint.h:109-145) */
static uint32_t widen_fp(uint32_t narrow, int frac_w, int exp_w) {
  const int bias = (1 << (exp_w - 1)) - 1;
  const uint32_t sign = (narrow >> (frac_w + exp_w)) & 1;
  const uint32_t ne = (narrow >> frac_w) & ((1u << exp_w) - 1);
  const uint32_t nf = narrow & ((1u << frac_w) - 1);
  uint32_t we = (uint32_t)((int32_t)ne - bias + 127);
  uint32_t wf = nf << (23 - frac_w);
  if (ne == ((1u << exp_w) - 1)) {
    we = 255; /* infinity or NaN; the fraction is kept / widened */
  } else if (ne == 0) {
    if (nf == 0) {
      we = 0;
      wf = 0;
    } else { /* subnormal: normalise */
      we = (uint32_t)(1 - bias + 127);
      while (!(wf & (1u << 23))) {
        we -= 1;
        wf <<= 1;
      }
      wf &= (1u << 23) - 1;
    }
  }
  return (sign << 31) | (we << 23) | wf;
}

/* constructor (:106-169) for an F32 image, then the dispatch of
 * readUncompressedRaw (:212-245) */
int oracle_unpack_f32_validate(const rsx_unpack_desc* d, const rsx_image* img,
                               size_t in_bytes) {
  const uint64_t need =
      (uint64_t)(uint32_t)d->crop_h * (uint64_t)(uint32_t)d->input_pitch_bytes;
  if (need > 0xFFFFFFFFull || need > in_bytes)
    return RSX_ERR_IO;
  if (d->crop_w <= 0 || d->crop_h <= 0)
    return RSX_ERR_INVALID_ARG;
  if (d->input_pitch_bytes < 1)
    return RSX_ERR_INVALID_ARG;
  if (d->bit_order < RSX_ORDER_LSB || d->bit_order > RSX_ORDER_MSB32)
    return RSX_ERR_INVALID_ARG;
  if (img->cpp < 1 || img->cpp > 3)
    return RSX_ERR_INVALID_ARG;
  if (d->bits_per_pixel < 1 || d->bits_per_pixel > 32)
    return RSX_ERR_INVALID_ARG; /* :138-140 */
  const uint64_t bits =
      (uint64_t)d->crop_w * (uint64_t)img->cpp * (uint64_t)d->bits_per_pixel;
  if (bits % 8 != 0)
    return RSX_ERR_INVALID_ARG;
  if ((uint64_t)d->input_pitch_bytes < bits / 8)
    return RSX_ERR_INVALID_ARG;
  if (d->crop_y < 0 || d->crop_x < 0)
    return RSX_ERR_INVALID_ARG;
  if ((uint64_t)d->crop_y > (uint64_t)img->dim_y)
    return RSX_ERR_INVALID_ARG;
  if ((uint64_t)d->crop_x + (uint64_t)d->crop_w > (uint64_t)img->dim_x)
    return RSX_ERR_INVALID_ARG;
  if (d->bits_per_pixel == 32)
    return RSX_OK; /* copyPixels :213-222 */
  if ((d->bit_order != RSX_ORDER_MSB && d->bit_order != RSX_ORDER_LSB) ||
      (d->bits_per_pixel != 16 && d->bits_per_pixel != 24))
    return RSX_ERR_INVALID_ARG; /* :243-244 */
  return RSX_OK;
}

int oracle_unpack_f32(const rsx_unpack_desc* d, const uint8_t* in, size_t in_bytes,
                      const rsx_image* img) {
  int st = oracle_unpack_f32_validate(d, img, in_bytes);
  if (st)
    return st;
  uint8_t* out = (uint8_t*)img->data;
  const int64_t cols = (int64_t)d->crop_w * img->cpp;
  int64_t y = d->crop_y;
  int64_t h = (int64_t)d->crop_h + d->crop_y;
  if (h > img->dim_y)
    h = img->dim_y;
  if (d->bits_per_pixel == 32) {
    for (int64_t r = y; r < h; ++r)
      memcpy(out + r * img->pitch_bytes + (size_t)d->crop_x * img->cpp * 4,
             in + (r - y) * d->input_pitch_bytes, (size_t)cols * 4);
    return RSX_OK;
  }
  /* decodePackedFP<Pump, Binary16 | Binary24> :171-186 */
  const int64_t stream_bytes = (int64_t)d->crop_h * d->input_pitch_bytes;
  const int64_t skip = d->input_pitch_bytes - cols * d->bits_per_pixel / 8;
  bitreader b;
  br_init(&b, in, stream_bytes, d->bit_order);
  if (b.err)
    return b.err;
  for (int64_t row = y; row < h; ++row) {
    uint32_t* orow = (uint32_t*)(out + row * img->pitch_bytes);
    for (int64_t col = 0; col < cols; ++col) {
      const uint32_t v = br_get(&b, d->bits_per_pixel);
      orow[d->crop_x + col] = d->bits_per_pixel == 16 ? widen_fp(v, 10, 5)
                                                      : widen_fp(v, 16, 7);
    }
    br_skip_bytes(&b, skip);
    if (b.err)
      return b.err;
  }
  return RSX_OK;
}

/* ---- the fixed-layout entry points of the same class ---------------------- */

/* UncompressedDecompressor::bytesPerLine (UncompressedDecompressor.cpp:88-104)
 * and the bpl the two sanityCheck(w, &h, bpp) callers use (:76-86). */
static int variant_bpl(const rsx_unpack_variant_desc* d, uint64_t* bpl) {
  const uint64_t w = (uint32_t)d->w;
  if (d->variant == RSX_UNPACK_8BIT_RAW || d->variant == RSX_UNPACK_8BIT_LOOKUP) {
    *bpl = w; /* sanityCheck(w, &h, 1) :273 */
  } else if (d->variant == RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED) {
    *bpl = 2 * w; /* sanityCheck(w, &h, 2) :360 */
  } else if (d->variant == RSX_UNPACK_12BIT_WITH_CONTROL) {
    if ((12 * w) % 8 != 0)
      return RSX_ERR_IO; /* "Bad image width" :91-92 */
    *bpl = 12 * w / 8 + (w + 2) / 10; /* :95-101 */
  } else {
    return RSX_ERR_INVALID_ARG;
  }
  return RSX_OK;
}

int oracle_unpack_variant_validate(const rsx_unpack_variant_desc* d,
                                   const rsx_image* img, size_t in_bytes) {
  if (d->variant < 0 || d->variant > RSX_UNPACK_8BIT_LOOKUP)
    return RSX_ERR_INVALID_ARG;
  if (d->w <= 0 || d->h <= 0)
    return RSX_ERR_INVALID_ARG; /* invariant(w > 0), invariant(*h > 0) */
  /* out(row, col) indexes the uncropped array: keep inside it */
  if (img->cpp < 1 || img->dim_x <= 0 || img->dim_y <= 0 ||
      (uint64_t)d->w > (uint64_t)img->dim_x * (uint64_t)img->cpp ||
      d->h > img->dim_y)
    return RSX_ERR_INVALID_ARG;
  uint64_t bpl = 0;
  int st = variant_bpl(d, &bpl);
  if (st)
    return st;
  /* sanityCheck(h, bpl) :52-70 then input.getData(bpl * h): IOException */
  if (bpl > 0x7FFFFFFFull || (uint64_t)in_bytes / bpl < (uint64_t)d->h)
    return RSX_ERR_IO;
  return RSX_OK;
}

int oracle_unpack_variant_u16(const rsx_unpack_variant_desc* d, const uint8_t* in,
                              size_t in_bytes, const rsx_image* img) {
  int st = oracle_unpack_variant_validate(d, img, in_bytes);
  if (st)
    return st;
  uint64_t bpl = 0;
  variant_bpl(d, &bpl);
  uint8_t* out = (uint8_t*)img->data;
  const uint32_t w = (uint32_t)d->w, h = (uint32_t)d->h;
  for (uint32_t row = 0; row < h; ++row) {
    uint16_t* o = (uint16_t*)(out + (size_t)row * img->pitch_bytes);
    const uint8_t* r = in + (size_t)row * bpl;
    if (d->variant == RSX_UNPACK_8BIT_RAW) {
      /* decode8BitRaw<true> :270-291 */
      for (uint32_t col = 0; col < w; ++col)
        o[col] = r[col];
    } else if (d->variant == RSX_UNPACK_8BIT_LOOKUP) {
      /* decode8BitRaw<false>: setWithLookUp with random == 0 throughout
       * (15700 * 0 + 0 == 0), so the dither term is (delta * 0 + 1024) >> 12 == 0 */
      for (uint32_t col = 0; col < w; ++col)
        o[col] = d->lut[r[col]];
    } else if (d->variant == RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED) {
      /* decode12BitRawUnpackedLeftAligned<e> :356-378 */
      for (uint32_t col = 0; col < w; ++col) {
        const uint32_t g1 = r[2 * col], g2 = r[2 * col + 1];
        const uint16_t pix =
            d->big_endian ? (uint16_t)((g1 << 8) | g2) : (uint16_t)((g2 << 8) | g1);
        o[col] = pix >> 4;
      }
    } else {
      /* decode12BitRawWithControl<e> :296-349: process(i, invert, p1, p2)
       * takes the "(p1 << 4) | (p2 >> 4)" form iff invert == (e == little) */
      uint32_t col = 0;
      for (uint32_t x = 0; x < w; x += 2) {
        uint32_t g1 = r[col], g2 = r[col + 1];
        if (d->big_endian)
          o[x] = (uint16_t)((g1 << 4) | (g2 >> 4));
        else
          o[x] = (uint16_t)(((g2 & 0x0f) << 8) | g1);
        g1 = r[col + 2];
        if (d->big_endian)
          o[x + 1] = (uint16_t)(((g2 & 0x0f) << 8) | g1);
        else
          o[x + 1] = (uint16_t)((g1 << 4) | (g2 >> 4));
        col += 3;
        if ((x % 10) == 8)
          col++;
      }
    }
  }
  return RSX_OK;
}

/* ======================================================================== */
/* Huffman: codes/HuffmanCode.h, PrefixCodeLookupDecoder.h,                    */
/* AbstractPrefixCodeDecoder.h                                                */
/* ======================================================================== */

typedef struct hufftab {
  int max_len;             /* maxCodeLength() */
  uint32_t max_code[17];   /* maxCodeOL, 0xFFFFFFFF = none (:105) */
  uint32_t code_offset[17]; /* codeOffsetOL */
  uint8_t values[RSX_MAX_CODE_VALUES];
  int n_values;
  int fix16;
} hufftab;

/* HuffmanCode::setNCodesPerLength / setCodeValues checks (HuffmanCode.h:99-166),
 * AbstractPrefixCodeTranscoder::verifyCodeValuesAsDiffLengths
 * (AbstractPrefixCodeTranscoder.h:71-84), generateCodeSymbols (:66-92) and
 * PrefixCodeLookupDecoder::setup (PrefixCodeLookupDecoder.h:97-113). */
static int huff_setup(hufftab* h, const rsx_huff_table* t) {
  int max_len = 16;
  while (max_len > 0 && t->n_codes_per_length[max_len - 1] == 0)
    --max_len;
  if (max_len == 0)
    return RSX_ERR_INVALID_ARG; /* "Codes-per-length table is empty" */
  unsigned count = 0;
  for (int l = 1; l <= max_len; ++l)
    count += t->n_codes_per_length[l - 1];
  if (count > RSX_MAX_CODE_VALUES || count != t->n_code_values)
    return RSX_ERR_INVALID_ARG;
  unsigned max_codes = 2;
  for (int l = 1; l <= max_len; ++l) {
    const unsigned n = t->n_codes_per_length[l - 1];
    if (n > (1u << l) || n > max_codes)
      return RSX_ERR_INVALID_ARG; /* "Corrupt Huffman" */
    max_codes -= n;
    max_codes *= 2;
  }
  for (unsigned i = 0; i < count; ++i)
    if (t->code_values[i] > 16)
      return RSX_ERR_INVALID_ARG; /* full-decode: value is a diff length */
  h->max_len = max_len;
  h->n_values = (int)count;
  h->fix16 = t->fix_dng_bug16 != 0;
  memcpy(h->values, t->code_values, count);
  uint32_t code = 0;
  unsigned so_far = 0;
  for (int l = 0; l <= 16; ++l) {
    h->max_code[l] = 0xFFFFFFFFu;
    h->code_offset[l] = 0xFFFFFFFFu;
  }
  for (int l = 1; l <= max_len; ++l) {
    const unsigned n = t->n_codes_per_length[l - 1];
    if (n) {
      h->code_offset[l] = (uint16_t)(code - so_far);
      h->max_code[l] = code + n - 1;
      so_far += n;
      code += n;
    }
    code <<= 1;
  }
  return RSX_OK;
}

/* One difference: PrefixCodeLookupDecoder::decode<_,true> (:183-193) =
 * fill(32); readSymbol (finishReadingPartialSymbol :133-164); processSymbol +
 * extend (AbstractPrefixCodeDecoder.h:43-76).  The reference's production
 * decoder (PrefixCodeLUTDecoder.h:172-216) consumes exactly the same bits and
 * fails on exactly the same inputs (fuzz/.../PrefixCodeDecoder/Dual.cpp). */
static int huff_decode_diff(const hufftab* h, bitreader* b, int* err) {
  br_fill(b, 32);
  if (b->err) {
    *err = b->err;
    return 0;
  }
  uint32_t code = 0;
  int len = 0;
  while (len < h->max_len &&
         (h->max_code[len] == 0xFFFFFFFFu || code > h->max_code[len])) {
    code = (code << 1) | br_get_nofill(b, 1);
    ++len;
  }
  if (len > h->max_len || h->max_code[len] == 0xFFFFFFFFu ||
      code > h->max_code[len]) {
    *err = RSX_ERR_BAD_HUFFMAN_CODE;
    return 0;
  }
  const unsigned idx = (code - h->code_offset[len]) & 0xFFFFu;
  const int ssss = h->values[idx];
  if (ssss == 0)
    return 0;
  if (ssss == 16) {
    if (h->fix16)
      br_skip_nofill(b, 16);
    return -32768;
  }
  const uint32_t v = br_get_nofill(b, ssss);
  int diff = (int)v;
  if ((v & (1u << (ssss - 1))) == 0)
    diff -= (1 << ssss) - 1;
  return diff;
}

/* Known-answer probe (test/librawspeed/codes/HuffmanTableTest.cpp:69-132):
 * decode `n` differences from a stream of the given bit order with one table
 * (the reference test drives the decoder with BitStreamerMSB). */
int oracle_huff_decode(const rsx_huff_table* t, int order, const uint8_t* data,
                       size_t size, int n, int32_t* out) {
  hufftab h;
  int st = huff_setup(&h, t);
  if (st)
    return st;
  bitreader b;
  br_init(&b, data, (int64_t)size, order);
  if (b.err)
    return b.err;
  for (int i = 0; i < n; ++i) {
    int err = 0;
    out[i] = huff_decode_diff(&h, &b, &err);
    if (err)
      return err;
  }
  return RSX_OK;
}

static int setup_tables(hufftab* tabs, const rsx_huff_table* src, int n_tables,
                        const uint8_t* table_index, int n_comp) {
  if (n_tables < 1 || n_tables > RSX_MAX_COMPONENTS)
    return RSX_ERR_INVALID_ARG;
  for (int i = 0; i < n_tables; ++i) {
    int st = huff_setup(&tabs[i], &src[i]);
    if (st)
      return st;
  }
  for (int c = 0; c < n_comp; ++c)
    if (table_index[c] >= n_tables)
      return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

/* ======================================================================== */
/* LJpegDecompressor                                                          */
/* ======================================================================== */

/* Constructor checks (LJpegDecompressor.cpp:52-152). */
int oracle_ljpeg_validate(const rsx_ljpeg_desc* d, const rsx_image* img,
                          size_t in_bytes) {
  (void)in_bytes;
  if (img->cpp < 1 || img->cpp > 3)
    return RSX_ERR_INVALID_ARG; /* :61-68 */
  if (img->dim_x <= 0 || img->dim_y <= 0)
    return RSX_ERR_INVALID_ARG; /* :70-71 */
  if (d->tile_w <= 0 || d->tile_h <= 0)
    return RSX_ERR_INVALID_ARG; /* :73-74 */
  if (d->tile_x < 0 || d->tile_y < 0)
    return RSX_ERR_INVALID_ARG;
  if (d->tile_x >= img->dim_x || d->tile_y >= img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :84-87 */
  if (d->tile_w > img->dim_x || d->tile_h > img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :89-92 */
  if ((int64_t)d->tile_x + d->tile_w > img->dim_x ||
      (int64_t)d->tile_y + d->tile_h > img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :94-97 */
  if (d->frame_w <= 0 || d->frame_h <= 0)
    return RSX_ERR_INVALID_ARG; /* :99-100 */
  const int mw = d->mcu_w, mh = d->mcu_h;
  if (!((mh == 1 && mw >= 1 && mw <= 4) || (mw == 2 && mh == 2)))
    return RSX_ERR_INVALID_ARG; /* :102-105 */
  if (d->n_comp != mw * mh)
    return RSX_ERR_INVALID_ARG; /* :107-108 */
  if (d->rows_per_restart_interval < 1)
    return RSX_ERR_INVALID_ARG; /* :115-116 */
  if ((int64_t)mw * d->frame_w > 0x7FFFFFFF ||
      (int64_t)mh * d->frame_h > 0x7FFFFFFF)
    return RSX_ERR_INVALID_ARG; /* :118-122 */
  if (d->tile_w < mw || d->tile_h < mh)
    return RSX_ERR_INVALID_ARG; /* :128-129 */
  if (d->tile_h % mh != 0)
    return RSX_ERR_INVALID_ARG; /* :131-132 */
  const int64_t req_w = (int64_t)img->cpp * d->tile_w;
  const int64_t mcus_to_consume = (req_w + mw - 1) / mw;
  if (d->frame_w < mcus_to_consume || (int64_t)mh * d->frame_h < d->tile_h ||
      (int64_t)mw * d->frame_w < req_w)
    return RSX_ERR_INVALID_ARG; /* :137-146 */
  hufftab tabs[RSX_MAX_COMPONENTS];
  return setup_tables(tabs, d->tables, d->n_tables, d->table_index, d->n_comp);
}

/* LJpegDecompressor::decode -> decodeN<MCU> (LJpegDecompressor.cpp:254-339)
 * -> decodeRowN (:184-251).  Writes uint16 samples into `img`; *consumed =
 * decodeN's return value (inputStream.getPosition(), :338). */
int oracle_ljpeg_decode(const rsx_ljpeg_desc* d, const uint8_t* in,
                        size_t in_bytes, const rsx_image* img,
                        uint32_t* consumed) {
  int st = oracle_ljpeg_validate(d, img, in_bytes);
  if (st)
    return st;
  hufftab tabs[RSX_MAX_COMPONENTS];
  setup_tables(tabs, d->tables, d->n_tables, d->table_index, d->n_comp);
  const int N = d->n_comp, mw = d->mcu_w, mh = d->mcu_h;
  const int64_t req_w = (int64_t)img->cpp * d->tile_w; /* img.width() :264-268 */
  const int64_t n_full = req_w / mw;                   /* :149 */
  const int trailing = (int)(req_w % mw);              /* :151 */
  uint8_t* base = (uint8_t*)img->data;
  const int64_t x0 = (int64_t)img->cpp * d->tile_x;

  const int64_t ljpeg_rows = d->tile_h / mh;
  const int64_t n_ri = (ljpeg_rows + d->rows_per_restart_interval - 1) /
                       d->rows_per_restart_interval; /* :277-278 */
  int64_t ipos = 0; /* inputStream position */
  for (int64_t ri = 0; ri < n_ri; ++ri) {
    uint16_t pred[4];
    for (int c = 0; c < N; ++c)
      pred[c] = d->init_pred[c]; /* :285-286 */
    if (ri != 0) { /* :288-298 */
      if (ipos + 2 > (int64_t)in_bytes)
        return RSX_ERR_IO; /* peekByte out of bounds */
      const uint8_t c0 = in[ipos], c1 = in[ipos + 1];
      if (!(c0 == 0xFF && c1 != 0 && c1 != 0xFF))
        return RSX_ERR_RESTART_MARKER; /* "Jpeg marker not encountered" */
      if (c1 < 0xD0 || c1 > 0xD7)
        return RSX_ERR_RESTART_MARKER; /* "Not a restart marker!" */
      if ((c1 - 0xD0) != ((ri - 1) % 8))
        return RSX_ERR_RESTART_MARKER; /* "Unexpected restart marker found" */
      ipos += 2;
    }
    bitreader b;
    br_init(&b, in + ipos, (int64_t)in_bytes - ipos, RSX_ORDER_JPEG); /* :300 */
    if (b.err)
      return b.err;
    for (int64_t rr = 0; rr < d->rows_per_restart_interval; ++rr) {
      const int64_t row = mh * (d->rows_per_restart_interval * ri + rr);
      if (row == d->tile_h)
        break; /* :311-315 */
      uint16_t* orow[2];
      for (int r = 0; r < mh; ++r)
        orow[r] = (uint16_t*)(base + (d->tile_y + row + r) * img->pitch_bytes) + x0;
      uint16_t first[4] = {0, 0, 0, 0};
      int64_t m = 0;
      int err = 0;
      for (; m < n_full; ++m) { /* :200-219 */
        for (int r = 0; r < mh; ++r)
          for (int cc = 0; cc < mw; ++cc) {
            const int c = mw * r + cc;
            const int diff = huff_decode_diff(&tabs[d->table_index[c]], &b, &err);
            if (err)
              return err;
            const uint16_t pix = (uint16_t)(pred[c] + diff);
            orow[r][mw * m + cc] = pix;
            pred[c] = pix; /* pred = outTile */
            if (m == 0)
              first[c] = pix;
          }
      }
      if (trailing != 0) { /* :222-244 */
        for (int r = 0; r < mh; ++r)
          for (int cc = 0; cc < mw; ++cc) {
            const int c = mw * r + cc;
            const int diff = huff_decode_diff(&tabs[d->table_index[c]], &b, &err);
            if (err)
              return err;
            const int64_t col = mw * m + cc;
            if (col < req_w)
              orow[r][col] = (uint16_t)(pred[c] + diff);
          }
        ++m;
      }
      for (; m < d->frame_w; ++m) /* discard the rest :247-250 */
        for (int c = 0; c < N; ++c) {
          huff_decode_diff(&tabs[d->table_index[c]], &b, &err);
          if (err)
            return err;
        }
      for (int c = 0; c < N; ++c)
        pred[c] = first[c]; /* next line predicts from start of this one :326-332 */
    }
    const int64_t sp = br_jpeg_stream_position(&b); /* :335 */
    if (ipos + sp > (int64_t)in_bytes)
      return RSX_ERR_IO; /* ByteStream::skipBytes bounds check */
    ipos += sp;
  }
  if (consumed)
    *consumed = (uint32_t)ipos;
  return RSX_OK;
}

/* ======================================================================== */
/* Cr2Decompressor                                                            */
/* ======================================================================== */

typedef struct rect {
  int x, y, w, h;
} rect;

typedef struct cr2geom {
  int N, xsf, ysf, sub, slice_col_step, px_per_group, group_size;
  int dim_x, dim_y;     /* in groups / rows  (Cr2DecompressorImpl.h:299-303) */
  int frame_x, frame_y; /* :305-311 */
  int n_slices, slice_w, last_w; /* in groups (:335-341) */
} cr2geom;

static int cr2_slice_width(const cr2geom* g, int id) {
  return id + 1 == g->n_slices ? g->last_w : g->slice_w;
}

/* Cr2OutputTileIterator (Cr2DecompressorImpl.h:104-154): pour slices
 * (width w_i, height frame_y) column-wise into the image.  Calls `cb` for
 * every tile of getAllOutputTiles(); stops early if cb returns non-zero. */
typedef int (*tile_cb)(void* ctx, rect r);
static int cr2_for_all_tiles(const cr2geom* g, tile_cb cb, void* ctx) {
  int ox = 0, oy = 0, slice_row = 0, id = 0;
  while (id < g->n_slices) {
    rect t = {ox, oy, cr2_slice_width(g, id), 0};
    int out_rem = g->dim_y - oy;
    int tile_rem = g->frame_y - slice_row;
    t.h = out_rem < tile_rem ? out_rem : tile_rem;
    int rc = cb(ctx, t);
    if (rc)
      return rc;
    slice_row += t.h;
    oy += t.h;
    if (slice_row == g->frame_y) {
      ++id;
      slice_row = 0;
    }
    if (oy == g->dim_y) {
      oy = 0;
      ox += t.w;
    }
  }
  return 0;
}

typedef struct cr2_validate_ctx {
  const cr2geom* g;
  int have_last;
  rect last;
  int result; /* 0 running, 1 stop ok, <0 error */
} cr2_validate_ctx;

/* evaluateConsecutiveTiles (Cr2DecompressorImpl.h:60-72). */
static int cr2_consecutive(rect a, rect b) {
  if (a.x == b.x && a.y + a.h == b.y && a.x + a.w == b.x + b.w)
    return 1; /* ContinuesColumn */
  if (b.y == 0 && b.x == a.x + a.w)
    return 2; /* BeginsNewColumn */
  return 0;   /* Invalid */
}

static int cr2_validate_cb(void* vctx, rect t) {
  cr2_validate_ctx* c = (cr2_validate_ctx*)vctx;
  const cr2geom* g = c->g;
  if (c->have_last && cr2_consecutive(c->last, t) == 0) {
    c->result = -1; /* "Invalid tiling" :348-350 */
    return 1;
  }
  if (t.x + t.w <= g->dim_x && t.y + t.h <= g->dim_y) {
    c->last = t;
    c->have_last = 1;
    return 0;
  }
  if (t.x < g->dim_x && t.y < g->dim_y) {
    c->result = -1; /* "Output tile partially outside of image" :355-356 */
    return 1;
  }
  c->result = 1; /* the rest do not contribute :357 */
  return 1;
}

static int cr2_geometry(const rsx_cr2_desc* d, const rsx_image* img,
                        cr2geom* g) {
  (void)img;
  /* format check :293-298 */
  const int N = d->n_comp, X = d->x_s_f, Y = d->y_s_f;
  if (!((N == 3 && X == 2 && Y == 2) || (N == 3 && X == 2 && Y == 1) ||
        (N == 2 && X == 1 && Y == 1) || (N == 4 && X == 1 && Y == 1)))
    return RSX_ERR_INVALID_ARG;
  g->N = N;
  g->xsf = X;
  g->ysf = Y;
  g->sub = (X != 1 || Y != 1);
  g->slice_col_step = N * X;
  g->px_per_group = X * Y;
  g->group_size = !g->sub ? N : 2 + g->px_per_group; /* Dsc :250-275 */
  return RSX_OK;
}

int oracle_cr2_validate(const rsx_cr2_desc* d, const rsx_image* img,
                        size_t in_bytes) {
  (void)in_bytes;
  cr2geom g;
  if (d->num_slices < 1)
    return RSX_ERR_INVALID_ARG; /* Cr2SliceWidths ctor, Cr2Decompressor.h:66-67 */
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :290-291 */
  int st = cr2_geometry(d, img, &g);
  if (st)
    return st;
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_x % g.group_size != 0)
    return RSX_ERR_INVALID_ARG; /* :300-302 */
  g.dim_x = img->dim_x / g.group_size;
  g.dim_y = img->dim_y;
  if (d->frame_w <= 0 || d->frame_h <= 0 || d->frame_w % g.xsf != 0 ||
      d->frame_h % g.ysf != 0)
    return RSX_ERR_INVALID_ARG; /* :305-308 */
  if (img->dim_x > 19440 || img->dim_y > 5920)
    return RSX_ERR_INVALID_ARG; /* :313-316 */
  /* widthOfSlice(i) > 0 for every slice :318-322 */
  for (int i = 0; i < d->num_slices; ++i) {
    const int w = i + 1 == d->num_slices ? d->last_slice_width : d->slice_width;
    if (w <= 0)
      return RSX_ERR_INVALID_ARG;
  }
  if (g.sub == (img->is_cfa != 0))
    return RSX_ERR_INVALID_ARG; /* :324-325 */
  /* rec.size() == N_COMP, full-decode tables :327-333 */
  hufftab tabs[RSX_MAX_COMPONENTS];
  st = setup_tables(tabs, d->tables, d->n_tables, d->table_index, d->n_comp);
  if (st)
    return st;
  if (d->slice_width % g.slice_col_step != 0 ||
      d->last_slice_width % g.slice_col_step != 0)
    return RSX_ERR_INVALID_ARG; /* :335-341 */
  g.frame_x = d->frame_w / g.xsf;
  g.frame_y = d->frame_h / g.ysf;
  g.n_slices = d->num_slices;
  g.slice_w = d->slice_width / g.slice_col_step;
  g.last_w = d->last_slice_width / g.slice_col_step;
  if ((int64_t)g.frame_x * g.frame_y < (int64_t)g.dim_x * g.dim_y)
    return RSX_ERR_INVALID_ARG; /* :343-344 */
  cr2_validate_ctx c = {&g, 0, {0, 0, 0, 0}, 0};
  cr2_for_all_tiles(&g, cr2_validate_cb, &c);
  if (c.result < 0)
    return RSX_ERR_INVALID_ARG;
  if (!c.have_last)
    return RSX_ERR_INVALID_ARG; /* "No tiles are provided" :359-360 */
  if (c.last.x + c.last.w != g.dim_x || c.last.y + c.last.h != g.dim_y)
    return RSX_ERR_INVALID_ARG; /* :361-362 */
  return RSX_OK;
}

typedef struct cr2_decode_ctx {
  const cr2geom* g;
  const rsx_cr2_desc* d;
  const hufftab* tabs;
  bitreader* b;
  uint8_t* base;
  uint32_t pitch;
  /* vertical strip being coalesced (Cr2VerticalOutputStripIterator :156-205) */
  int have_strip;
  rect strip;
  int done; /* reached the tile whose bottom-right == dim (getOutputTiles) */
  /* decode state (Cr2DecompressorImpl.h:410-428) */
  uint16_t pred[4];
  int pn_row, pn_col; /* predNext = out[pn_row] block at group pn_col */
  int frame_col;
  int err;
} cr2_decode_ctx;

static uint16_t* cr2_px(cr2_decode_ctx* c, int row, int sample) {
  return (uint16_t*)(c->base + (int64_t)row * c->pitch) + sample;
}

/* The three nested loops of decompressN_X_Y over one vertical output strip
 * (Cr2DecompressorImpl.h:431-465). */
static void cr2_decode_strip(cr2_decode_ctx* c, rect out) {
  const cr2geom* g = c->g;
  const int gs = g->group_size;
  for (int row = out.y; row != out.y + out.h; ++row) {
    for (int col = out.x; col != out.x + out.w;) {
      if (g->frame_x - c->frame_col == 0) { /* :437-451 */
        for (int k = 0; k < g->N; ++k) {
          const int idx = k == 0 ? 0 : gs - (g->N - k);
          c->pred[k] = *cr2_px(c, c->pn_row, gs * c->pn_col + idx);
        }
        c->pn_row = row;
        c->pn_col = col;
        c->frame_col = 0;
      }
      int end = col + (g->frame_x - c->frame_col);
      if (end > out.x + out.w)
        end = out.x + out.w;
      for (; col != end; ++col, ++c->frame_col) { /* :455-463 */
        for (int p = 0; p < gs; ++p) {
          const int k = p < g->px_per_group ? 0 : p - g->px_per_group + 1;
          const int diff = huff_decode_diff(&c->tabs[c->d->table_index[k]],
                                            c->b, &c->err);
          if (c->err)
            return;
          c->pred[k] = (uint16_t)(c->pred[k] + diff);
          *cr2_px(c, row, gs * col + p) = c->pred[k];
        }
      }
    }
  }
}

static int cr2_decode_cb(void* vctx, rect t) {
  cr2_decode_ctx* c = (cr2_decode_ctx*)vctx;
  const cr2geom* g = c->g;
  if (c->done)
    return 1;
  if (c->have_strip) {
    if (cr2_consecutive(c->strip, t) == 1) {
      c->strip.h += t.h; /* coalesce :170-187 */
    } else {
      cr2_decode_strip(c, c->strip);
      if (c->err)
        return 1;
      c->strip = t;
    }
  } else {
    c->strip = t;
    c->have_strip = 1;
  }
  if (t.x + t.w == g->dim_x && t.y + t.h == g->dim_y)
    c->done = 1; /* getOutputTiles :224-231 */
  return 0;
}

/* Cr2Decompressor::decompress -> decompressN_X_Y (Cr2DecompressorImpl.h:396-485). */
int oracle_cr2_decode(const rsx_cr2_desc* d, const uint8_t* in, size_t in_bytes,
                      const rsx_image* img, uint32_t* consumed) {
  int st = oracle_cr2_validate(d, img, in_bytes);
  if (st)
    return st;
  cr2geom g;
  cr2_geometry(d, img, &g);
  g.dim_x = img->dim_x / g.group_size;
  g.dim_y = img->dim_y;
  g.frame_x = d->frame_w / g.xsf;
  g.frame_y = d->frame_h / g.ysf;
  g.n_slices = d->num_slices;
  g.slice_w = d->slice_width / g.slice_col_step;
  g.last_w = d->last_slice_width / g.slice_col_step;
  hufftab tabs[RSX_MAX_COMPONENTS];
  setup_tables(tabs, d->tables, d->n_tables, d->table_index, d->n_comp);
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_JPEG); /* :419 */
  if (b.err)
    return b.err;
  cr2_decode_ctx c;
  memset(&c, 0, sizeof c);
  c.g = &g;
  c.d = d;
  c.tabs = tabs;
  c.b = &b;
  c.base = (uint8_t*)img->data;
  c.pitch = img->pitch_bytes;
  for (int k = 0; k < g.N; ++k)
    c.pred[k] = d->init_pred[k];
  cr2_for_all_tiles(&g, cr2_decode_cb, &c);
  if (!c.err && c.have_strip)
    cr2_decode_strip(&c, c.strip);
  if (c.err)
    return c.err;
  if (consumed)
    *consumed = (uint32_t)br_jpeg_stream_position(&b); /* :467 */
  return RSX_OK;
}

/* ======================================================================== */
/* NikonDecompressor (decompressors/NikonDecompressor.cpp)                    */
/* ======================================================================== */

/* NikonLASDecompressor (.cpp:79-377): JPEG Annex C/F tables of the "lossy
 * after split" trees.  The 14-bit bigTable (:225-294) is an accelerator with
 * the same results as this slow path for every entry it accepts (plain SSSS
 * values with code + SSSS <= 14 bits); everything else falls through to it. */
typedef struct lastab {
  int bits[17];
  int huffval[256];
  int mincode[17];
  int maxcode[18];
  int valptr[17];
  int numbits[256];
} lastab;

static int las_validate(const rsx_huff_table* t) {
  unsigned total = 0;
  uint32_t code = 0;
  for (int l = 1; l <= 16; ++l) {
    const unsigned n = t->n_codes_per_length[l - 1];
    total += n;
    code += n;
    if (code > (1u << l))
      return RSX_ERR_INVALID_ARG;
    code <<= 1;
  }
  if (total == 0 || total > RSX_MAX_CODE_VALUES || total != t->n_code_values)
    return RSX_ERR_INVALID_ARG;
  /* values the reference cannot decode without undefined behaviour
   * (getBits(len - shl) with a negative count, :373) are rejected */
  for (unsigned i = 0; i < total; ++i) {
    const unsigned v = t->code_values[i];
    if (v != 16 && (v >> 4) != 0 && (v >> 4) >= (v & 15u))
      return RSX_ERR_INVALID_ARG; /* getBits(<= 0) */
  }
  return RSX_OK;
}

/* createPrefixCodeDecoder :108-213 */
static void las_setup(lastab* t, const rsx_huff_table* h) {
  int huffsize[258], huffcode[258];
  int p = 0;
  memset(t, 0, sizeof *t);
  for (int l = 1; l <= 16; ++l) {
    t->bits[l] = h->n_codes_per_length[l - 1];
    for (int i = 1; i <= t->bits[l]; ++i)
      huffsize[p++] = l; /* Figure C.1 */
  }
  huffsize[p] = 0;
  const int lastp = p;
  for (int i = 0; i < h->n_code_values; ++i)
    t->huffval[i] = h->code_values[i];
  int code = 0, si = huffsize[0];
  p = 0;
  while (huffsize[p]) { /* Figure C.2 */
    while (huffsize[p] == si)
      huffcode[p++] = code++;
    code <<= 1;
    si++;
  }
  p = 0;
  for (int l = 1; l <= 16; ++l) { /* Figure F.15 */
    if (t->bits[l]) {
      t->valptr[l] = p;
      t->mincode[l] = huffcode[p];
      p += t->bits[l];
      t->maxcode[l] = huffcode[p - 1];
    } else {
      t->valptr[l] = 0xff;
      t->maxcode[l] = -1;
    }
  }
  t->maxcode[17] = 0xFFFFF;
  for (p = 0; p < lastp; ++p) { /* :188-206 */
    const int size = huffsize[p];
    if (size <= 8) {
      const int ll = huffcode[p] << (8 - size);
      const int ul = size < 8 ? (ll | ((1 << (8 - size)) - 1)) : ll;
      for (int i = ll; i <= ul && i < 256; ++i)
        t->numbits[i] = size | (t->huffval[p] << 4);
    }
  }
}

/* decodeDifference :331-376 */
static int las_decode_diff(const lastab* t, bitreader* b, int* err) {
  br_fill(b, 32);
  if (b->err) {
    *err = b->err;
    return 0;
  }
  int rv;
  int code = (int)br_peek_nofill(b, 8);
  const int val = t->numbits[code];
  int l = val & 15;
  if (l) {
    br_skip_nofill(b, l);
    rv = val >> 4;
  } else {
    br_skip_nofill(b, 8);
    l = 8;
    while (code > t->maxcode[l]) {
      code = (code << 1) | (int)br_get_nofill(b, 1);
      l++;
    }
    if (l > 16) {
      *err = RSX_ERR_BAD_HUFFMAN_CODE; /* "Corrupt JPEG data: bad Huffman code" */
      return 0;
    }
    rv = t->huffval[t->valptr[l] + (code - t->mincode[l])];
  }
  if (rv == 16)
    return -32768;
  const int len = rv & 15, shl = rv >> 4;
  if (len == 0)
    return 0; /* bigTable entry "rv == 0" :290-291 */
  const int nb = len - shl;
  const uint32_t bits = nb ? br_get(b, nb) : 0;
  if (b->err) {
    *err = b->err;
    return 0;
  }
  int diff = (int)((((bits << 1) + 1) << shl) >> 1);
  if ((diff & (1 << (len - 1))) == 0)
    diff -= (1 << len) - !shl;
  return diff;
}

int oracle_nikon_validate(const rsx_nikon_desc* d, const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :476-478 */
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_x % 2 != 0 ||
      img->dim_x > 8288 || img->dim_y > 5520)
    return RSX_ERR_INVALID_ARG; /* :480-483 */
  if (d->bits_ps != 12 && d->bits_ps != 14)
    return RSX_ERR_INVALID_ARG; /* :485-491 */
  if (d->split < 0 || d->split >= img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :511-512: out-of-image splits arrive as 0 */
  for (int i = 0; i < 4; ++i)
    if ((&d->p_up[0][0])[i] < 0 || (&d->p_up[0][0])[i] > 65535)
      return RSX_ERR_INVALID_ARG; /* getU16 :506-509 */
  if (!d->uncorrected_raw_values &&
      (d->curve == NULL || d->curve_size < 1 || d->curve_size > 65536))
    return RSX_ERR_INVALID_ARG; /* TableLookUp.cpp:50-57 */
  hufftab h;
  int st = huff_setup(&h, &d->tables[0]);
  if (st)
    return st;
  if (d->tables[0].fix_dng_bug16)
    return RSX_ERR_INVALID_ARG; /* ht.setup(true, false) :468 */
  if (d->split != 0 && (st = las_validate(&d->tables[1])))
    return st;
  return RSX_OK;
}

/* decompress(input, uncorrectedRawValues) :541-560 and
 * decompress<Huffman>(bits, start_y, end_y) :515-539, with
 * RawImageDataU16::setWithLookUp (common/RawImage.h:335-353) and
 * TableLookUp::setTable (common/TableLookUp.cpp:50-84) inlined. */
int oracle_nikon_decompress(const rsx_nikon_desc* d, const uint8_t* in,
                            size_t in_bytes, const rsx_image* img) {
  int st = oracle_nikon_validate(d, img);
  if (st)
    return st;
  /* the dithering table (RawImageCurveGuard -> setTable(curve, true)) */
  uint16_t* tab = NULL;
  if (!d->uncorrected_raw_values) {
    tab = (uint16_t*)malloc(2 * 65536 * sizeof(uint16_t));
    const int n = d->curve_size;
    for (int i = 0; i < 65536; ++i) {
      if (i < n) {
        const int center = d->curve[i];
        int lower = i > 0 ? d->curve[i - 1] : center;
        int upper = i < n - 1 ? d->curve[i + 1] : center;
        if (lower > center)
          lower = center;
        if (upper < center)
          upper = center;
        const int delta = upper - lower;
        int base = center - ((upper - lower + 2) / 4);
        base = base < 0 ? 0 : (base > 65535 ? 65535 : base);
        tab[2 * i] = (uint16_t)base;
        tab[2 * i + 1] = (uint16_t)delta;
      } else {
        tab[2 * i] = d->curve[n - 1];
        tab[2 * i + 1] = 0;
      }
    }
  }
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_MSB);
  if (b.err) {
    free(tab);
    return b.err;
  }
  br_fill(&b, 24);
  uint32_t random = br_peek_nofill(&b, 24); /* :549 */
  hufftab h0;
  lastab h1;
  huff_setup(&h0, &d->tables[0]);
  if (d->split)
    las_setup(&h1, &d->tables[1]);
  int pup[2][2] = {{d->p_up[0][0], d->p_up[0][1]}, {d->p_up[1][0], d->p_up[1][1]}};
  const int W = img->dim_x, H = img->dim_y;
  int err = 0;
  for (int row = 0; row < H && !err; ++row) {
    const int after = d->split && row >= d->split;
    int pred[2] = {pup[row & 1][0], pup[row & 1][1]};
    uint16_t* o = (uint16_t*)((uint8_t*)img->data + (size_t)row * img->pitch_bytes);
    for (int col = 0; col < W; ++col) {
      pred[col & 1] += after ? las_decode_diff(&h1, &b, &err)
                             : huff_decode_diff(&h0, &b, &err);
      if (err)
        break;
      if (col < 2)
        pup[row & 1][col & 1] = pred[col & 1];
      int v = pred[col & 1];
      v = v < 0 ? 0 : (v > 32767 ? 32767 : v); /* clampBits(.., 15) */
      if (!tab) {
        o[col] = (uint16_t)v;
      } else {
        const uint32_t base = tab[2 * v], delta = tab[2 * v + 1];
        o[col] = (uint16_t)(base + ((delta * (random & 2047) + 1024) >> 12));
        random = 15700 * (random & 65535) + (random >> 16);
      }
    }
  }
  free(tab);
  return err;
}

/* ======================================================================== */
/* PentaxDecompressor (decompressors/PentaxDecompressor.cpp)                  */
/* ======================================================================== */

int oracle_pentax_validate(const rsx_pentax_desc* d, const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :58-60 */
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_x % 2 != 0 ||
      img->dim_x > 8384 || img->dim_y > 6208)
    return RSX_ERR_INVALID_ARG; /* :62-66 */
  hufftab h;
  int st = huff_setup(&h, &d->table);
  if (st)
    return st;
  if (d->table.fix_dng_bug16)
    return RSX_ERR_INVALID_ARG; /* ht.setup(true, false) :147 */
  return RSX_OK;
}

/* decompress :152-176 */
int oracle_pentax_decompress(const rsx_pentax_desc* d, const uint8_t* in,
                             size_t in_bytes, const rsx_image* img) {
  int st = oracle_pentax_validate(d, img);
  if (st)
    return st;
  hufftab h;
  huff_setup(&h, &d->table);
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_MSB);
  if (b.err)
    return b.err;
  const int W = img->dim_x, H = img->dim_y;
  int err = 0;
  for (int row = 0; row < H; ++row) {
    uint16_t* o = (uint16_t*)((uint8_t*)img->data + (size_t)row * img->pitch_bytes);
    int pred[2] = {0, 0};
    if (row >= 2) {
      const uint16_t* up =
          (const uint16_t*)((const uint8_t*)img->data + (size_t)(row - 2) * img->pitch_bytes);
      pred[0] = up[0];
      pred[1] = up[1];
    }
    for (int col = 0; col < W; ++col) {
      pred[col & 1] += huff_decode_diff(&h, &b, &err);
      if (err)
        return err;
      const int value = pred[col & 1];
      if (((unsigned)value >> 16) != 0)
        return RSX_ERR_VALUE_RANGE; /* !isIntN(value, 16) :170-171 */
      o[col] = (uint16_t)value;
    }
  }
  return RSX_OK;
}

/* ======================================================================== */
/* SamsungV1Decompressor (decompressors/SamsungV1Decompressor.cpp)            */
/* ======================================================================== */

int oracle_samsung_v1_validate(const rsx_samsung_v1_desc* d, const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :48-50 */
  if (d->bits != 12)
    return RSX_ERR_INVALID_ARG; /* :53-54 */
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_x % 32 != 0 ||
      img->dim_y % 2 != 0 || img->dim_x > 5664 || img->dim_y > 3714)
    return RSX_ERR_INVALID_ARG; /* :59-61 */
  if (d->n_entries < 1 || d->n_entries > RSX_SAMSUNG_V1_MAX_ENTRIES)
    return RSX_ERR_INVALID_ARG;
  uint32_t filled = 0;
  for (int i = 0; i < d->n_entries; ++i) {
    if (d->enc_len[i] < 1 || d->enc_len[i] > 10 || d->diff_len[i] > 13)
      return RSX_ERR_INVALID_ARG;
    filled += 1024u >> d->enc_len[i];
  }
  return filled == 1024 ? RSX_OK : RSX_ERR_INVALID_ARG; /* the table is 1024 entries :102 */
}

/* decompress :81-140 with samsungDiff :63-79 */
int oracle_samsung_v1_decompress(const rsx_samsung_v1_desc* d, const uint8_t* in,
                                 size_t in_bytes, const rsx_image* img) {
  int st = oracle_samsung_v1_validate(d, img);
  if (st)
    return st;
  uint8_t enc[1024], dif[1024];
  uint32_t n = 0;
  for (int i = 0; i < d->n_entries; ++i) /* :110-117 */
    for (uint32_t c = 0; c < (1024u >> d->enc_len[i]); ++c) {
      enc[n] = d->enc_len[i];
      dif[n] = d->diff_len[i];
      ++n;
    }
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_MSB);
  if (b.err)
    return b.err;
  const int W = img->dim_x, H = img->dim_y;
  for (int row = 0; row < H; ++row) {
    uint16_t* o = (uint16_t*)((uint8_t*)img->data + (size_t)row * img->pitch_bytes);
    int pred[2] = {0, 0};
    if (row >= 2) {
      const uint16_t* up =
          (const uint16_t*)((const uint8_t*)img->data + (size_t)(row - 2) * img->pitch_bytes);
      pred[0] = up[0];
      pred[1] = up[1];
    }
    for (int col = 0; col < W; ++col) {
      br_fill(&b, 23); /* :66 */
      if (b.err)
        return b.err;
      const uint32_t c = br_peek_nofill(&b, 10);
      br_skip_nofill(&b, enc[c]);
      const int len = dif[c];
      int diff = 0;
      if (len) {
        const uint32_t v = br_get_nofill(&b, len);
        diff = (int)v;
        if ((v & (1u << (len - 1))) == 0)
          diff -= (1 << len) - 1; /* PrefixCodeDecoder<>::extend */
      }
      pred[col & 1] += diff;
      const int value = pred[col & 1];
      if (((unsigned)value >> d->bits) != 0)
        return RSX_ERR_VALUE_RANGE; /* !isIntN(value, bits) :133-134 */
      o[col] = (uint16_t)value;
    }
  }
  return RSX_OK;
}

/* ======================================================================== */
/* SonyArw1Decompressor (decompressors/SonyArw1Decompressor.cpp)              */
/* ======================================================================== */

int oracle_sony_arw1_validate(const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :41-43 */
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_y % 2 != 0 || img->dim_x > 4600 ||
      img->dim_y > 3072)
    return RSX_ERR_INVALID_ARG; /* :48-49 */
  return RSX_OK;
}

/* decompress :59-93 */
int oracle_sony_arw1_decompress(const uint8_t* in, size_t in_bytes, const rsx_image* img) {
  int st = oracle_sony_arw1_validate(img);
  if (st)
    return st;
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_MSB);
  if (b.err)
    return b.err;
  const int W = img->dim_x, H = img->dim_y;
  int pred = 0;
  for (int col = W - 1; col >= 0; --col) {
    for (int row = 0; row < H + 1; row += 2) {
      br_fill(&b, 32); /* :70 */
      if (b.err)
        return b.err;
      if (row == H)
        row = 1; /* :72-73: the odd rows follow the even ones */
      uint32_t len = 4 - br_get_nofill(&b, 2);
      if (len == 3 && br_get_nofill(&b, 1))
        len = 0;
      if (len == 4)
        while (len < 17 && !br_get_nofill(&b, 1))
          len++;
      int diff = 0;
      if (len) { /* getDiff :53-57 */
        const uint32_t v = br_get_nofill(&b, (int)len);
        diff = (int)v;
        if ((v & (1u << (len - 1))) == 0)
          diff -= (1 << len) - 1; /* PrefixCodeDecoder<>::extend */
      }
      pred += diff;
      if (((unsigned)pred >> 12) != 0)
        return RSX_ERR_VALUE_RANGE; /* !isIntN(pred, 12) :88-89: adt/Bit.h:85-90 tests
                                       the value AS UNSIGNED, i.e. 0 <= pred < 4096 */
      ((uint16_t*)((uint8_t*)img->data + (size_t)row * img->pitch_bytes))[col] =
          (uint16_t)pred;
    }
  }
  return RSX_OK;
}

/* ======================================================================== */
/* Cr2sRawInterpolator (interpolators/Cr2sRawInterpolator.cpp)                */
/* ======================================================================== */

int oracle_sraw_validate(const rsx_sraw_desc* d, const rsx_image* in,
                         const rsx_image* out) {
  if (d->version < 0 || d->version > 2)
    return RSX_ERR_INVALID_ARG; /* :511 */
  if (d->subsampling_y != 1 && d->subsampling_y != 2)
    return RSX_ERR_INVALID_ARG; /* :540-541 */
  if (d->subsampling_y == 2 && d->version == 0)
    return RSX_ERR_INVALID_ARG; /* :529-538 */
  const int gs = 2 + 2 * d->subsampling_y;
  if (in->cpp != 1 || out->cpp != 3)
    return RSX_ERR_INVALID_ARG;
  if (in->dim_x <= 0 || in->dim_y <= 0 || in->dim_x % gs != 0)
    return RSX_ERR_INVALID_ARG; /* :104, :201 */
  const int mcus = in->dim_x / gs;
  if (mcus <= 1)
    return RSX_ERR_INVALID_ARG; /* :106, :203 */
  if (out->dim_x != 2 * mcus || out->dim_y != d->subsampling_y * in->dim_y)
    return RSX_ERR_INVALID_ARG; /* Cr2Decoder.cpp:589-596 */
  return RSX_OK;
}

typedef struct ycc {
  int Y, Cb, Cr;
} ycc;

static uint16_t clamp16(int v) { return (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v)); }

/* YUV_TO_RGB<version> :470-506 + STORE_RGB :462-468 (int arithmetic; the
 * products are computed unsigned so that the wrap is defined) */
static void sraw_store(const rsx_sraw_desc* d, ycc p, uint16_t* o) {
  int r, g, b;
  if (d->version == 0) {
    r = p.Y + p.Cr - 512;
    g = p.Y + ((-778 * p.Cb - (p.Cr * 2048)) >> 12) - 512;
    b = p.Y + (p.Cb - 512);
  } else if (d->version == 1) {
    r = p.Y + ((50 * p.Cb + 22929 * p.Cr) >> 12);
    g = p.Y + ((-5640 * p.Cb - 11751 * p.Cr) >> 12);
    b = p.Y + ((29040 * p.Cb - 101 * p.Cr) >> 12);
  } else {
    r = p.Y + p.Cr;
    g = p.Y + ((-778 * p.Cb - (p.Cr * 2048)) >> 12);
    b = p.Y + p.Cb;
  }
  r = (int)((uint32_t)d->sraw_coeffs[0] * (uint32_t)r);
  g = (int)((uint32_t)d->sraw_coeffs[1] * (uint32_t)g);
  b = (int)((uint32_t)d->sraw_coeffs[2] * (uint32_t)b);
  o[0] = clamp16(r >> 8);
  o[1] = clamp16(g >> 8);
  o[2] = clamp16(b >> 8);
}

/* LoadMCU (:113-122 / :208-223): the Ys, and the chroma into pixel [0] */
static void sraw_load(const uint16_t* row, int m, int gs, ycc* px /* gs - 2 pixels */) {
  for (int i = 0; i < gs - 2; ++i) {
    px[i].Y = row[gs * m + i];
    px[i].Cb = px[i].Cr = 0;
  }
  px[0].Cb = row[gs * m + gs - 2];
  px[0].Cr = row[gs * m + gs - 1];
}
static void sraw_process(ycc* p, int hue) { /* signExtend + applyHue :69-80 */
  p->Cb += hue - 16384;
  p->Cr += hue - 16384;
}

int oracle_sraw_interpolate(const rsx_sraw_desc* d, const rsx_image* in,
                            const rsx_image* out) {
  int st = oracle_sraw_validate(d, in, out);
  if (st)
    return st;
  const int gs = 2 + 2 * d->subsampling_y, n = in->dim_x / gs, H = in->dim_y;
  const uint8_t* ib = (const uint8_t*)in->data;
  uint8_t* ob = (uint8_t*)out->data;
#define IROW(r) ((const uint16_t*)(ib + (size_t)(r) * in->pitch_bytes))
#define OROW(r) ((uint16_t*)(ob + (size_t)(r) * out->pitch_bytes))
  if (gs == 4) { /* interpolate_422_row :95-176 */
    for (int row = 0; row < H; ++row) {
      int m;
      for (m = 0; m < n - 1; ++m) {
        ycc a[2], b[2];
        sraw_load(IROW(row), m, 4, a);
        sraw_load(IROW(row), m + 1, 4, b);
        sraw_process(&a[0], d->hue);
        sraw_process(&b[0], d->hue);
        a[1].Cb = (a[0].Cb + b[0].Cb) >> 1;
        a[1].Cr = (a[0].Cr + b[0].Cr) >> 1;
        sraw_store(d, a[0], OROW(row) + 6 * m);
        sraw_store(d, a[1], OROW(row) + 6 * m + 3);
      }
      ycc a[2];
      sraw_load(IROW(row), m, 4, a);
      sraw_process(&a[0], d->hue);
      a[1].Cb = a[0].Cb;
      a[1].Cr = a[0].Cr;
      sraw_store(d, a[0], OROW(row) + 6 * m);
      sraw_store(d, a[1], OROW(row) + 6 * m + 3);
    }
    return RSX_OK;
  }
  /* interpolate_420: rows 0 .. H-2 by interpolate_420_row (:188-345), then the last
   * input row (:385-460); pixel [MCURow][MCUCol] = px[2 * MCURow + MCUCol] */
  for (int row = 0; row < H - 1; ++row) {
    int m;
    for (m = 0; m < n - 1; ++m) {
      ycc a[4], b[4], c[4], e[4]; /* (row,m) (row,m+1) (row+1,m) (row+1,m+1) */
      sraw_load(IROW(row), m, 6, a);
      sraw_load(IROW(row), m + 1, 6, b);
      sraw_load(IROW(row + 1), m, 6, c);
      sraw_load(IROW(row + 1), m + 1, 6, e);
      sraw_process(&a[0], d->hue);
      sraw_process(&b[0], d->hue);
      sraw_process(&c[0], d->hue);
      sraw_process(&e[0], d->hue);
      a[1].Cb = (a[0].Cb + b[0].Cb) >> 1;
      a[1].Cr = (a[0].Cr + b[0].Cr) >> 1;
      a[2].Cb = (a[0].Cb + c[0].Cb) >> 1;
      a[2].Cr = (a[0].Cr + c[0].Cr) >> 1;
      a[3].Cb = (a[0].Cb + b[0].Cb + c[0].Cb + e[0].Cb) >> 2;
      a[3].Cr = (a[0].Cr + b[0].Cr + c[0].Cr + e[0].Cr) >> 2;
      for (int k = 0; k < 4; ++k)
        sraw_store(d, a[k], OROW(2 * row + (k >> 1)) + 6 * m + 3 * (k & 1));
    }
    ycc a[4], c[4]; /* last MCU of the line :333-345 */
    sraw_load(IROW(row), m, 6, a);
    sraw_load(IROW(row + 1), m, 6, c);
    sraw_process(&a[0], d->hue);
    sraw_process(&c[0], d->hue);
    a[2].Cb = (a[0].Cb + c[0].Cb) >> 1;
    a[2].Cr = (a[0].Cr + c[0].Cr) >> 1;
    a[1].Cb = a[0].Cb;
    a[1].Cr = a[0].Cr;
    a[3].Cb = a[2].Cb;
    a[3].Cr = a[2].Cr;
    for (int k = 0; k < 4; ++k)
      sraw_store(d, a[k], OROW(2 * row + (k >> 1)) + 6 * m + 3 * (k & 1));
  }
  {
    const int row = H - 1;
    int m;
    for (m = 0; m < n - 1; ++m) { /* :397-423 */
      ycc a[4], b[4];
      sraw_load(IROW(row), m, 6, a);
      sraw_load(IROW(row), m + 1, 6, b);
      sraw_process(&a[0], d->hue);
      sraw_process(&b[0], d->hue);
      a[1].Cb = (a[0].Cb + b[0].Cb) >> 1;
      a[1].Cr = (a[0].Cr + b[0].Cr) >> 1;
      a[2].Cb = a[0].Cb;
      a[2].Cr = a[0].Cr;
      a[3].Cb = a[1].Cb;
      a[3].Cr = a[1].Cr;
      for (int k = 0; k < 4; ++k)
        sraw_store(d, a[k], OROW(2 * row + (k >> 1)) + 6 * m + 3 * (k & 1));
    }
    ycc a[4]; /* :440-459 */
    sraw_load(IROW(row), m, 6, a);
    sraw_process(&a[0], d->hue);
    for (int k = 1; k < 4; ++k) {
      a[k].Cb = a[0].Cb;
      a[k].Cr = a[0].Cr;
    }
    for (int k = 0; k < 4; ++k)
      sraw_store(d, a[k], OROW(2 * row + (k >> 1)) + 6 * m + 3 * (k & 1));
  }
#undef IROW
#undef OROW
  return RSX_OK;
}

/* ======================================================================== */
/* HasselbladDecompressor (decompressors/HasselbladDecompressor.cpp)          */
/* ======================================================================== */

int oracle_hasselblad_validate(const rsx_hasselblad_desc* d, const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :44-45 */
  if (img->dim_x <= 0 || img->dim_y <= 0 || img->dim_x % 2 != 0 ||
      img->dim_x > 12000 || img->dim_y > 8842)
    return RSX_ERR_INVALID_ARG; /* :48-52 */
  hufftab h;
  int st = huff_setup(&h, &d->table); /* incl. verifyCodeValuesAsDiffLengths :80 */
  if (st)
    return st;
  if (d->table.fix_dng_bug16)
    return RSX_ERR_INVALID_ARG;
  return RSX_OK;
}

/* PrefixCodeLookupDecoder::decodeCodeValue: fill(32), then the code walk */
static int huff_decode_value(const hufftab* h, bitreader* b, int* err) {
  br_fill(b, 32);
  if (b->err) {
    *err = b->err;
    return 0;
  }
  uint32_t code = 0;
  int len = 0;
  while (len < h->max_len &&
         (h->max_code[len] == 0xFFFFFFFFu || code > h->max_code[len])) {
    code = (code << 1) | br_get_nofill(b, 1);
    ++len;
  }
  if (len > h->max_len || h->max_code[len] == 0xFFFFFFFFu || code > h->max_code[len]) {
    *err = RSX_ERR_BAD_HUFFMAN_CODE;
    return 0;
  }
  return h->values[(code - h->code_offset[len]) & 0xFFFFu];
}

/* getBits :60-69 */
static int hb_get_bits(bitreader* b, int len, int* err) {
  if (!len)
    return 0;
  const uint32_t v = br_get(b, len);
  if (b->err) {
    *err = b->err;
    return 0;
  }
  int diff = (int)v;
  if ((v & (1u << (len - 1))) == 0)
    diff -= (1 << len) - 1; /* PrefixCodeDecoder<>::extend */
  if (diff == 65535)
    return -32768;
  return diff;
}

/* decompress :71-100; *consumed = bitStreamer.getStreamPosition() =
 * getInputPosition() - (fillLevel >> 3) (BitStreamer.h:238-241) */
int oracle_hasselblad_decompress(const rsx_hasselblad_desc* d, const uint8_t* in,
                                 size_t in_bytes, const rsx_image* img,
                                 uint32_t* consumed) {
  int st = oracle_hasselblad_validate(d, img);
  if (st)
    return st;
  hufftab h;
  huff_setup(&h, &d->table);
  bitreader b;
  br_init(&b, in, (int64_t)in_bytes, RSX_ORDER_MSB32);
  if (b.err)
    return b.err;
  int err = 0;
  for (int row = 0; row < img->dim_y; ++row) {
    uint16_t* o = (uint16_t*)((uint8_t*)img->data + (size_t)row * img->pitch_bytes);
    int p1 = d->init_pred, p2 = d->init_pred;
    for (int col = 0; col < img->dim_x; col += 2) {
      const int len1 = huff_decode_value(&h, &b, &err);
      if (err)
        return err;
      const int len2 = huff_decode_value(&h, &b, &err);
      if (err)
        return err;
      p1 += hb_get_bits(&b, len1, &err);
      if (err)
        return err;
      p2 += hb_get_bits(&b, len2, &err);
      if (err)
        return err;
      o[col] = (uint16_t)p1;
      o[col + 1] = (uint16_t)p2;
    }
  }
  if (consumed)
    *consumed = (uint32_t)(b.pos - (b.fill >> 3));
  return RSX_OK;
}


/* ======================================================================== */
/* SamsungV2Decompressor (decompressors/SamsungV2Decompressor.cpp)           */
/* Not yet served by the GPU library: the restatement is here  */
/* so that the kernel of a later round has its checker; pinned against the   */
/* reference build in tests/test_oracle_samsung_v2.py.                       */
/* ======================================================================== */

typedef struct sv2 {
  const rsx_image* img;
  int bits, width, height;
  unsigned optflags; /* 1 SKIP, 2 MV, 4 QP  (:46-54) */
  int init_val;
  int motion, scale;
  int mode[3][2]; /* diffBitsMode */
} sv2;

static uint16_t* sv2_px(const sv2* s, int row, int col) {
  return (uint16_t*)((uint8_t*)s->img->data + (size_t)row * s->img->pitch_bytes) + col;
}

/* prepareBaselineValues :152-230 */
static int sv2_baseline(sv2* s, bitreader* b, int row, int col, uint16_t base[16]) {
  if (!(s->optflags & 4u) && (col % 64) == 0) { /* :159-164 */
    static const int scalevals[3] = {0, -2, 2};
    const uint32_t i = br_get(b, 2);
    s->scale = i < 3 ? s->scale + scalevals[i] : (int)br_get(b, 12);
  }
  if (s->optflags & 2u) /* :167-170 */
    s->motion = br_get(b, 1) ? 3 : 7;
  else if (!br_get(b, 1))
    s->motion = (int)br_get(b, 3);
  if (b->err)
    return b->err;
  if ((row == 0 || row == 1) && s->motion != 7)
    return RSX_ERR_INVALID_ARG; /* :172-173 */
  if (s->motion == 7) { /* :175-188 */
    if (col == 0) {
      for (int i = 0; i < 16; ++i)
        base[i] = (uint16_t)s->init_val;
      return RSX_OK;
    }
    for (int i = 0; i < 16; ++i)
      base[i] = *sv2_px(s, row, col + (i & 1) - 2);
    return RSX_OK;
  }
  if (row < 2)
    return RSX_ERR_INVALID_ARG; /* :191-192 */
  static const int motion_offset[7] = {-4, -2, -2, 0, 0, 2, 4};
  static const int motion_avg[7] = {0, 0, 1, 0, 1, 0, 0};
  const int slide = motion_offset[s->motion], avg = motion_avg[s->motion];
  for (int i = 0; i < 16; ++i) { /* :202-227 */
    int ref_row = row, ref_col = col + i + slide;
    if ((row + i) & 1) {
      ref_row -= 2;
    } else {
      ref_row -= 1;
      ref_col += (i & 1) ? -1 : 1;
    }
    if (ref_col < 0)
      return RSX_ERR_INVALID_ARG;
    if (ref_col >= s->width || (avg && ref_col + 2 >= s->width))
      return RSX_ERR_INVALID_ARG;
    if (avg)
      base[i] = (uint16_t)((*sv2_px(s, ref_row, ref_col) + *sv2_px(s, ref_row, ref_col + 2) + 1) >> 1);
    else
      base[i] = *sv2_px(s, ref_row, ref_col);
  }
  return RSX_OK;
}

/* decodeDiffLengths :232-277 + decodeDifferences :279-314 */
static int sv2_differences(sv2* s, bitreader* b, int row, int scaled[16]) {
  uint32_t diff_bits[4] = {0, 0, 0, 0};
  if ((s->optflags & 1u) || !br_get(b, 1)) { /* :234 (SKIP: always coded) */
    uint32_t flags[4];
    for (int i = 0; i < 4; ++i)
      flags[i] = br_get(b, 2);
    for (int i = 0; i < 4; ++i) {
      const int colornum = (row % 2 != 0) ? i >> 1 : ((i >> 1) + 2) % 3;
      switch (flags[i]) {
      case 0:
        diff_bits[i] = (uint32_t)s->mode[colornum][0];
        break;
      case 1:
        diff_bits[i] = (uint32_t)s->mode[colornum][0] + 1;
        break;
      case 2:
        if (s->mode[colornum][0] == 0)
          return RSX_ERR_INVALID_ARG; /* :258-259 */
        diff_bits[i] = (uint32_t)s->mode[colornum][0] - 1;
        break;
      default:
        diff_bits[i] = br_get(b, 4);
        break;
      }
      s->mode[colornum][0] = s->mode[colornum][1];
      s->mode[colornum][1] = (int)diff_bits[i];
      if (diff_bits[i] > (uint32_t)s->bits + 1)
        return RSX_ERR_INVALID_ARG; /* :271-272 */
    }
  }
  if (b->err)
    return b->err;
  int16_t diffs[16], shuffled[16];
  for (int i = 0; i < 16; ++i) { /* :286-290 with getDiff :79-85 */
    const uint32_t len = diff_bits[i >> 2];
    int v = 0;
    if (len) {
      const uint32_t u = br_get(b, (int)len);
      v = (int)(u << (32 - len)) >> (32 - len); /* signExtend */
    }
    diffs[i] = (int16_t)v;
  }
  if (b->err)
    return b->err;
  for (int i = 0; i < 16; ++i) { /* :293-304 */
    const int p = (row % 2) ? ((i % 8) << 1) - (i >> 3) + 1 : ((i % 8) << 1) + (i >> 3);
    shuffled[p] = diffs[i];
  }
  for (int i = 0; i < 16; ++i) /* :307-311 */
    scaled[i] = (int)shuffled[i] * (s->scale * 2 + 1) + s->scale;
  return RSX_OK;
}

/* The constructor's header parse (:87-141) and decompress / decompressRow (:312-338).
 * `in` = the strip the container hands over (header included). */
int oracle_samsung_v2_decompress(const uint8_t* in, size_t in_bytes, int bits,
                                 const rsx_image* img) {
  if (img->cpp != 1)
    return RSX_ERR_INVALID_ARG; /* :90-92 */
  if (bits != 12 && bits != 14)
    return RSX_ERR_INVALID_ARG; /* :94-100 */
  if (in_bytes < 16)
    return RSX_ERR_IO; /* bs.check(headerSize) :103 */
  bitreader h;
  br_init(&h, in, (int64_t)in_bytes, RSX_ORDER_MSB32);
  sv2 s;
  memset(&s, 0, sizeof s);
  s.img = img;
  br_get(&h, 16);
  br_get(&h, 4);
  s.bits = (int)br_get(&h, 4) + 1;
  if (s.bits != bits)
    return RSX_ERR_INVALID_ARG; /* :112-113 */
  br_get(&h, 4);
  br_get(&h, 4);
  s.width = (int)br_get(&h, 16);
  s.height = (int)br_get(&h, 16);
  br_get(&h, 16);
  br_get(&h, 4);
  s.optflags = br_get(&h, 4);
  if (s.optflags > 7u)
    return RSX_ERR_INVALID_ARG; /* :123-125 */
  br_get(&h, 8);
  br_get(&h, 8);
  br_get(&h, 8);
  br_get(&h, 2);
  s.init_val = (int)br_get(&h, 14);
  if (s.width == 0 || s.height == 0 || s.width % 16 != 0 || s.width > 6496 || s.height > 4336)
    return RSX_ERR_INVALID_ARG; /* :134-136 */
  if (s.width != img->dim_x || s.height != img->dim_y)
    return RSX_ERR_INVALID_ARG; /* :138-139 */
  const uint8_t* data = in + 16;
  const int64_t size = (int64_t)in_bytes - 16;
  int64_t pos = 0;
  for (int row = 0; row < s.height; ++row) {
    if (pos & 0xf) { /* :314-316 */
      const int64_t skip = 16 - (pos & 0xf);
      if (pos + skip > size)
        return RSX_ERR_IO;
      pos += skip;
    }
    bitreader b;
    br_init(&b, data + pos, size - pos, RSX_ORDER_MSB32);
    if (b.err)
      return b.err;
    s.motion = 7;
    s.scale = 0;
    for (int c = 0; c < 3; ++c)
      s.mode[c][0] = s.mode[c][1] = (row == 0 || row == 1) ? 7 : 4;
    for (int col = 0; col < s.width; col += 16) { /* processBlock :316-330 */
      uint16_t base[16];
      int scaled[16];
      int st = sv2_baseline(&s, &b, row, col, base);
      if (st)
        return st;
      st = sv2_differences(&s, &b, row, scaled);
      if (st)
        return st;
      for (int i = 0; i < 16; ++i) {
        int v = (int)base[i] + scaled[i];
        const int hi = (1 << s.bits) - 1;
        v = v < 0 ? 0 : (v > hi ? hi : v); /* clampBits */
        *sv2_px(&s, row, col + i) = (uint16_t)v;
      }
    }
    const int64_t used = b.pos - (b.fill >> 3); /* getStreamPosition */
    if (pos + used > size)
      return RSX_ERR_IO;
    pos += used;
  }
  return RSX_OK;
}
