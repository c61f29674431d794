/*
 * rsx_synth.c -- host-side stream writers used to synthesise inputs.
 *
 * The reference ships writers for every bit order and a Huffman encoder that
 * only its tests / fuzzers / benchmarks use (bitstreams/BitVacuumer*.h,
 * codes/PrefixCodeVectorEncoder.h:80-90).  This file is our own equivalent:
 * a packed-integer row writer for the four UncompressedDecompressor bit
 * orders, a lossless-JPEG (predictor 1) scan + container writer that emits
 * exactly the layout AbstractLJpegDecoder accepts, and
 * a deterministic "sensor-like" image model.  Plain C, no GPU, no reference
 * code.  Built into rawspeed_amd/librsx_synth.so.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Packed-integer rows (the inverse of decodePackedInt,                      */
/* UncompressedDecompressor.cpp:188-200; stream addressing SURVEY.md A.1).   */
/* ------------------------------------------------------------------------ */

/* Writes rows x cols samples of `bps` bits each, every row followed by zero
 * padding up to `pitch_bytes`, as ONE continuous bit stream in the given
 * order (0 LSB, 1 MSB, 2 MSB16, 3 MSB32).  Returns bytes written
 * (rows*pitch_bytes) or 0 on a bad argument (pitch too small, bits not a
 * multiple of 8, or total size not a multiple of the order's word size). */
size_t rsx_synth_pack_rows(int order, int bps, const uint16_t* samples,
                           size_t sample_stride, int cols, int rows,
                           int pitch_bytes, uint8_t* out) {
  const uint64_t row_bits = (uint64_t)cols * (uint64_t)bps;
  if (bps < 1 || bps > 16 || row_bits % 8 != 0 ||
      (uint64_t)pitch_bytes < row_bits / 8 || order < 0 || order > 3)
    return 0;
  const size_t total = (size_t)rows * (size_t)pitch_bytes;
  const size_t word = order == 2 ? 2 : (order == 3 ? 4 : 1);
  if (total % word != 0)
    return 0;
  memset(out, 0, total);
  const uint32_t mask = (1u << bps) - 1u;
  for (int r = 0; r < rows; ++r) {
    uint8_t* row = out + (size_t)r * pitch_bytes;
    const uint16_t* src = samples + (size_t)r * sample_stride;
    uint64_t acc = 0;
    int nacc = 0;
    size_t o = 0;
    if (order == 0) { /* LSB first */
      for (int x = 0; x < cols; ++x) {
        acc |= (uint64_t)(src[x] & mask) << nacc;
        nacc += bps;
        while (nacc >= 8) {
          row[o++] = (uint8_t)acc;
          acc >>= 8;
          nacc -= 8;
        }
      }
    } else { /* MSB first byte stream; word swizzle applied afterwards */
      for (int x = 0; x < cols; ++x) {
        acc = (acc << bps) | (src[x] & mask);
        nacc += bps;
        while (nacc >= 8) {
          row[o++] = (uint8_t)(acc >> (nacc - 8));
          nacc -= 8;
        }
      }
    }
  }
  if (order == 2) { /* MSB16: stream byte 2i <-> memory byte 2i+1 */
    for (size_t i = 0; i + 1 < total; i += 2) {
      uint8_t t = out[i];
      out[i] = out[i + 1];
      out[i + 1] = t;
    }
  } else if (order == 3) { /* MSB32: reverse within LE u32 words */
    for (size_t i = 0; i + 3 < total; i += 4) {
      uint8_t a = out[i], b = out[i + 1];
      out[i] = out[i + 3];
      out[i + 1] = out[i + 2];
      out[i + 2] = b;
      out[i + 3] = a;
    }
  }
  return total;
}

/* ------------------------------------------------------------------------ */
/* Deterministic pixel sources                                               */
/* ------------------------------------------------------------------------ */

static uint64_t splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* Uniform random `bits`-bit samples. */
void rsx_synth_uniform(uint16_t* out, size_t n, int bits, uint64_t seed) {
  uint64_t s = seed * 0x2545F4914F6CDD1Dull + 1;
  const uint16_t mask = (uint16_t)((1u << bits) - 1u);
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    uint64_t r = splitmix64(&s);
    out[i] = (uint16_t)r & mask;
    out[i + 1] = (uint16_t)(r >> 16) & mask;
    out[i + 2] = (uint16_t)(r >> 32) & mask;
    out[i + 3] = (uint16_t)(r >> 48) & mask;
  }
  if (i < n) {
    uint64_t r = splitmix64(&s);
    for (; i < n; ++i, r >>= 16)
      out[i] = (uint16_t)r & mask;
  }
}

/* "Sensor-like" w x h image (SURVEY.md 8(d) cfg 3): smooth ramp + 2x2 CFA
 * offset + approximately Gaussian noise (sum of four uniforms, sigma ~= 24),
 * clamped to [0, 2^prec - 1]. */
void rsx_synth_sensor_image(uint16_t* out, int w, int h, size_t stride,
                            int prec, uint64_t seed) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  const int maxv = (1 << prec) - 1;
  /* sum of 4 U[0,65536): mean 131070, sigma = 65536*sqrt(4/12) = 37837.2 */
  const double k = 24.0 / 37837.2;
  for (int y = 0; y < h; ++y) {
    uint16_t* row = out + (size_t)y * stride;
    for (int x = 0; x < w; ++x) {
      uint64_t r = splitmix64(&s);
      int sum = (int)(r & 0xFFFF) + (int)((r >> 16) & 0xFFFF) +
                (int)((r >> 32) & 0xFFFF) + (int)(r >> 48);
      double base = 2000.0 + 6000.0 * x / w + 3000.0 * y / h +
                    500.0 * ((x ^ y) & 1);
      int v = (int)(base + (sum - 131070) * k + 0.5);
      if (v < 0)
        v = 0;
      if (v > maxv)
        v = maxv;
      row[x] = (uint16_t)v;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Lossless JPEG writer                                                      */
/* ------------------------------------------------------------------------ */

typedef struct jpeg_writer {
  uint8_t* out;
  size_t cap, n;
  uint64_t acc;
  int nacc;
  int overflow;
  int raw; /* 1: plain MSB bit stream, no FF00 stuffing (BitStreamerMSB input) */
} jpeg_writer;

static void jw_byte_raw(jpeg_writer* w, uint8_t b) {
  if (w->n < w->cap)
    w->out[w->n] = b;
  else
    w->overflow = 1;
  w->n++;
}
static void jw_data_byte(jpeg_writer* w, uint8_t b) {
  jw_byte_raw(w, b);
  if (b == 0xFF && !w->raw)
    jw_byte_raw(w, 0x00); /* byte stuffing */
}
static void jw_bits(jpeg_writer* w, uint32_t v, int n) {
  if (n == 0)
    return;
  w->acc = (w->acc << n) | (v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u)));
  w->nacc += n;
  while (w->nacc >= 8) {
    jw_data_byte(w, (uint8_t)(w->acc >> (w->nacc - 8)));
    w->nacc -= 8;
  }
}
/* pad with 1-bits to a byte boundary */
static void jw_flush(jpeg_writer* w) {
  if (w->nacc > 0)
    jw_bits(w, 0xFFu, 8 - w->nacc);
  w->acc = 0;
}

typedef struct enc_table {
  uint16_t code[17];
  uint8_t len[17]; /* 0 = category absent from the table */
} enc_table;

/* canonical code from DHT counts + values (JPEG Annex C) */
static int enc_table_build(enc_table* t, const uint8_t counts[16],
                           const uint8_t* values, int n_values) {
  memset(t, 0, sizeof *t);
  uint32_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < counts[l - 1]; ++i, ++k) {
      if (k >= n_values || values[k] > 16)
        return -1;
      t->code[values[k]] = (uint16_t)code;
      t->len[values[k]] = (uint8_t)l;
      code++;
    }
    code <<= 1;
  }
  return k == n_values ? 0 : -1;
}

/* Encodes `rows` stream-order rows of `row_samples` samples (component of
 * sample s = s % n_comp; row_samples = frame_w * n_comp) with predictor 1 as
 * LJpegDecompressor / Cr2Decompressor reconstruct it:
 *   pred(r, s) = s >= n_comp ? X[r][s - n_comp]
 *              : (r is the first row of a restart interval ? init_pred[s]
 *                                                          : X[r-1][s]).
 * tables: n_comp pointers to {16 counts, values} pairs (may alias).
 * rows_per_ri: 0 = no restart markers; else RSTn (n = (i-1)%8) is written
 * between intervals.  fix16: emit 16 extra (zero) bits after an SSSS=16 code.
 * Returns the number of entropy bytes, or 0 on overflow / missing category. */
size_t rsx_synth_ljpeg_encode_pattern(const uint16_t* samples, size_t row_stride,
                                      int row_samples, int rows, int n_comp,
                                      int period, const uint8_t* comp_of_phase,
                                      const uint16_t* init_pred,
                                      const uint8_t* const* counts,
                                      const uint8_t* const* values,
                                      const int* n_values, int rows_per_ri,
                                      int fix16, uint8_t* out, size_t cap,
                                      uint64_t* n_symbol_bits);

size_t rsx_synth_ljpeg_encode_scan(const uint16_t* samples, size_t row_stride,
                                   int row_samples, int rows, int n_comp,
                                   const uint16_t* init_pred,
                                   const uint8_t* const* counts,
                                   const uint8_t* const* values,
                                   const int* n_values, int rows_per_ri,
                                   int fix16, uint8_t* out, size_t cap,
                                   uint64_t* n_symbol_bits) {
  static const uint8_t identity[4] = {0, 1, 2, 3};
  return rsx_synth_ljpeg_encode_pattern(samples, row_stride, row_samples, rows, n_comp,
                                        n_comp, identity, init_pred, counts, values,
                                        n_values, rows_per_ri, fix16, out, cap,
                                        n_symbol_bits);
}

/* General form: sample s of a row belongs to component comp_of_phase[s % period]
 * (period <= 8).  Its predictor is the previous sample of the same component in
 * the row; for the first one of a row it is init_pred[c] (first row / first row
 * of a restart interval) or the first sample of that component in the row above.
 * period == n_comp with the identity map is plain LJPEG predictor 1; period 4
 * (0,0,1,2) / 6 (0,0,0,0,1,2) are Canon sRaw <3,2,1> / <3,2,2>
 * (the layout Cr2DecompressorImpl.h:431-465 decodes). */
size_t rsx_synth_ljpeg_encode_pattern(const uint16_t* samples, size_t row_stride,
                                      int row_samples, int rows, int n_comp,
                                      int period, const uint8_t* comp_of_phase,
                                      const uint16_t* init_pred,
                                      const uint8_t* const* counts,
                                      const uint8_t* const* values,
                                      const int* n_values, int rows_per_ri,
                                      int fix16, uint8_t* out, size_t cap,
                                      uint64_t* n_symbol_bits) {
  enc_table tabs[4];
  if (n_comp < 1 || n_comp > 4 || period < 1 || period > 8)
    return 0;
  for (int c = 0; c < n_comp; ++c)
    if (enc_table_build(&tabs[c], counts[c], values[c], n_values[c]))
      return 0;
  int first_pos[4] = {-1, -1, -1, -1};
  for (int p = period - 1; p >= 0; --p) {
    if (comp_of_phase[p] >= n_comp)
      return 0;
    first_pos[comp_of_phase[p]] = p;
  }
  jpeg_writer w = {out, cap, 0, 0, 0, 0, 0};
  uint64_t bits = 0;
  for (int r = 0; r < rows; ++r) {
    const uint16_t* cur = samples + (size_t)r * row_stride;
    const uint16_t* up = r > 0 ? samples + (size_t)(r - 1) * row_stride : NULL;
    int ri_first = rows_per_ri > 0 ? (r % rows_per_ri == 0) : (r == 0);
    if (rows_per_ri > 0 && r > 0 && ri_first) {
      jw_flush(&w);
      jw_byte_raw(&w, 0xFF);
      jw_byte_raw(&w, (uint8_t)(0xD0 + ((r / rows_per_ri - 1) % 8)));
    }
    int last[4] = {-1, -1, -1, -1};
    for (int s = 0; s < row_samples; ++s) {
      const int c = comp_of_phase[s % period];
      uint16_t pred;
      if (last[c] >= 0)
        pred = cur[last[c]];
      else
        pred = ri_first ? init_pred[c] : up[first_pos[c]];
      last[c] = s;
      const int d = (int16_t)(uint16_t)(cur[s] - pred);
      int ssss = 0;
      if (d == -32768) {
        ssss = 16;
      } else {
        int a = d < 0 ? -d : d;
        while (a) {
          ++ssss;
          a >>= 1;
        }
      }
      const enc_table* t = &tabs[c];
      if (t->len[ssss] == 0)
        return 0; /* category not in table */
      jw_bits(&w, t->code[ssss], t->len[ssss]);
      bits += t->len[ssss];
      if (ssss == 16) {
        if (fix16) {
          jw_bits(&w, 0, 16);
          bits += 16;
        }
      } else if (ssss) {
        const uint32_t v = d >= 0 ? (uint32_t)d : (uint32_t)(d + (1 << ssss) - 1);
        jw_bits(&w, v, ssss);
        bits += ssss;
      }
    }
  }
  jw_flush(&w);
  if (n_symbol_bits)
    *n_symbol_bits = bits;
  return w.overflow ? 0 : w.n;
}

/* NikonDecompressor stream (NikonDecompressor.cpp:515-539) for an image of
 * 15-bit values: plain MSB bit stream, one Huffman table, the sample at
 * (row, col) is predicted from (row, col - 2), the first two of a row from
 * (row - 2, col) and those of rows 0 / 1 from p_up[2 * row + col].  Differences
 * are plain ints here (the decoder does not wrap), so |diff| must fit the
 * table's categories.  Returns the byte count (padded with zero bits), 0 on
 * overflow / missing category. */
static size_t nikon_encode_tab(const uint16_t* samples, size_t row_stride, int w, int h,
                               const int32_t* p_up, const enc_table* tab, uint8_t* out,
                               size_t cap, uint64_t* n_symbol_bits) {
  jpeg_writer wr = {out, cap, 0, 0, 0, 0, 1};
  uint64_t bits = 0;
  for (int r = 0; r < h; ++r) {
    const uint16_t* cur = samples + (size_t)r * row_stride;
    for (int x = 0; x < w; ++x) {
      int pred;
      if (x >= 2)
        pred = cur[x - 2];
      else if (r >= 2)
        pred = samples[(size_t)(r - 2) * row_stride + x];
      else
        pred = p_up[2 * r + x];
      const int d = (int)cur[x] - pred;
      int ssss = 0;
      for (int a = d < 0 ? -d : d; a; a >>= 1)
        ++ssss;
      if (ssss > 15 || tab->len[ssss] == 0)
        return 0;
      jw_bits(&wr, tab->code[ssss], tab->len[ssss]);
      bits += tab->len[ssss] + ssss;
      if (ssss)
        jw_bits(&wr, d >= 0 ? (uint32_t)d : (uint32_t)(d + (1 << ssss) - 1), ssss);
    }
  }
  if (wr.nacc > 0)
    jw_bits(&wr, 0, 8 - wr.nacc);
  if (n_symbol_bits)
    *n_symbol_bits = bits;
  return wr.overflow ? 0 : wr.n;
}

size_t rsx_synth_nikon_encode(const uint16_t* samples, size_t row_stride, int w,
                              int h, const int32_t* p_up, const uint8_t* counts,
                              const uint8_t* values, int n_values, uint8_t* out,
                              size_t cap, uint64_t* n_symbol_bits) {
  enc_table tab;
  if (enc_table_build(&tab, counts, values, n_values))
    return 0;
  return nikon_encode_tab(samples, row_stride, w, h, p_up, &tab, out, cap, n_symbol_bits);
}

/* Same stream layout with a prefix code that is not a canonical JPEG one
 * (SamsungV1Decompressor.cpp:88-117): entry i = (enc_len[i], diff_len[i]) owns
 * the next 1024 >> enc_len[i] slots of a 10-bit table, so its code is the first
 * slot's index >> (10 - enc_len[i]). */
size_t rsx_synth_prefix_encode(const uint16_t* samples, size_t row_stride, int w, int h,
                               const int32_t* p_up, const uint8_t* enc_len,
                               const uint8_t* diff_len, int n_entries, uint8_t* out,
                               size_t cap, uint64_t* n_symbol_bits) {
  enc_table tab;
  memset(&tab, 0, sizeof tab);
  unsigned pos = 0;
  for (int i = 0; i < n_entries; ++i) {
    if (enc_len[i] < 1 || enc_len[i] > 10 || diff_len[i] > 16)
      return 0;
    tab.code[diff_len[i]] = (uint16_t)(pos >> (10 - enc_len[i]));
    tab.len[diff_len[i]] = enc_len[i];
    pos += 1024u >> enc_len[i];
  }
  if (pos != 1024)
    return 0;
  return nikon_encode_tab(samples, row_stride, w, h, p_up, &tab, out, cap, n_symbol_bits);
}

/* SonyArw1Decompressor stream (SonyArw1Decompressor.cpp:59-93): plain MSB-first
 * bits; per pixel the length code -- "11" 1, "10" 2, "010" 3, "011" 0, "00" +
 * k zeros + "1" = 4 + k (the final "1" is absent for 17) -- then the difference
 * bits with JPEG sign extension.  One predictor (start 0) runs over the whole
 * image in decode order: columns right to left, in a column the even rows and
 * then the odd rows.  `samples` may hold any int16-representable value stored as
 * uint16 (the decoder rejects values outside 0..4095). */
size_t rsx_synth_sony_arw1_encode(const uint16_t* samples, size_t row_stride, int w, int h,
                                  uint8_t* out, size_t cap, uint64_t* n_symbol_bits) {
  jpeg_writer wr = {out, cap, 0, 0, 0, 0, 1};
  uint64_t bits = 0;
  int pred = 0;
  if (h & 1)
    return 0;
  for (int col = w - 1; col >= 0; --col) {
    for (int i = 0; i < h; ++i) {
      const int row = i < h / 2 ? 2 * i : 2 * (i - h / 2) + 1;
      const int cur = (int16_t)samples[(size_t)row * row_stride + col];
      const int d = cur - pred;
      pred = cur;
      int len = 0;
      for (int a = d < 0 ? -d : d; a; a >>= 1)
        ++len;
      if (len > 17)
        return 0;
      int nb;
      if (len == 0) {
        jw_bits(&wr, 3, 3); /* 011 */
        nb = 3;
      } else if (len <= 3) {
        static const uint8_t code[4] = {0, 3, 2, 2}, clen[4] = {0, 2, 2, 3};
        jw_bits(&wr, code[len], clen[len]); /* 11, 10, 010 */
        nb = clen[len];
      } else {
        const int k = len - 4;
        jw_bits(&wr, 0, 2);
        if (k)
          jw_bits(&wr, 0, k);
        nb = 2 + k;
        if (len < 17) {
          jw_bits(&wr, 1, 1);
          ++nb;
        }
      }
      bits += (uint64_t)(nb + len);
      if (len)
        jw_bits(&wr, d >= 0 ? (uint32_t)d : (uint32_t)(d + (1 << len) - 1), len);
    }
  }
  if (wr.nacc > 0)
    jw_bits(&wr, 0, 8 - wr.nacc);
  if (n_symbol_bits)
    *n_symbol_bits = bits;
  return wr.overflow ? 0 : wr.n;
}

/* HasselbladDecompressor stream (HasselbladDecompressor.cpp:71-100): pixels two at
 * a time as [len1 code][len2 code][len1 bits][len2 bits], both predictors restart
 * from init_pred on every row, arithmetic mod 2^16; BitStreamerMSB32 input, i.e.
 * the MSB-first bit stream is stored as little-endian 32-bit words.  Returns the
 * byte count (a multiple of 4), 0 on overflow / missing category. */
size_t rsx_synth_hasselblad_encode(const uint16_t* samples, size_t row_stride, int w,
                                   int h, unsigned init_pred, const uint8_t* counts,
                                   const uint8_t* values, int n_values, uint8_t* out,
                                   size_t cap, uint64_t* n_symbol_bits) {
  enc_table tab;
  if (enc_table_build(&tab, counts, values, n_values) || (w & 1))
    return 0;
  jpeg_writer wr = {out, cap, 0, 0, 0, 0, 1};
  uint64_t bits = 0;
  for (int r = 0; r < h; ++r) {
    const uint16_t* cur = samples + (size_t)r * row_stride;
    uint16_t p[2] = {(uint16_t)init_pred, (uint16_t)init_pred};
    for (int x = 0; x < w; x += 2) {
      int d[2], ssss[2];
      for (int k = 0; k < 2; ++k) {
        d[k] = (int16_t)(uint16_t)(cur[x + k] - p[k]); /* -32768 .. 32767 */
        p[k] = cur[x + k];
        ssss[k] = 0;
        if (d[k] == -32768) {
          ssss[k] = 16; /* the all-ones field */
        } else {
          for (int a = d[k] < 0 ? -d[k] : d[k]; a; a >>= 1)
            ++ssss[k];
        }
        if (tab.len[ssss[k]] == 0)
          return 0;
      }
      for (int k = 0; k < 2; ++k) {
        jw_bits(&wr, tab.code[ssss[k]], tab.len[ssss[k]]);
        bits += tab.len[ssss[k]] + ssss[k];
      }
      for (int k = 0; k < 2; ++k) {
        if (ssss[k] == 16)
          jw_bits(&wr, 0xFFFFu, 16);
        else if (ssss[k])
          jw_bits(&wr, d[k] >= 0 ? (uint32_t)d[k] : (uint32_t)(d[k] + (1 << ssss[k]) - 1),
                  ssss[k]);
      }
    }
  }
  if (wr.nacc > 0)
    jw_bits(&wr, 0, 8 - wr.nacc);
  while (wr.n % 4)
    jw_byte_raw(&wr, 0);
  if (wr.overflow)
    return 0;
  for (size_t i = 0; i + 4 <= wr.n; i += 4) { /* MSB-first words -> little-endian */
    uint8_t t = out[i];
    out[i] = out[i + 3];
    out[i + 3] = t;
    t = out[i + 1];
    out[i + 1] = out[i + 2];
    out[i + 2] = t;
  }
  if (n_symbol_bits)
    *n_symbol_bits = bits;
  return wr.n;
}

static void put16(uint8_t** p, unsigned v) {
  *(*p)++ = (uint8_t)(v >> 8);
  *(*p)++ = (uint8_t)v;
}

/* Writes SOI, SOF3, one DHT per distinct table, [DRI], SOS into `out`
 * (all fields big-endian,
 * AbstractLJpegDecoder.cpp:127-228).  comp_table[c] = DHT slot (0..3) of
 * component c; slot_counts/slot_values/slot_n describe slot i for i <
 * n_slots.  h_samp/v_samp give the SOF HiVi nibbles (1/1 for raw data).
 * Returns header size. */
size_t rsx_synth_ljpeg_header(uint8_t* out, int prec, int frame_w, int frame_h,
                              int n_comp, const int* comp_table, int n_slots,
                              const uint8_t* const* slot_counts,
                              const uint8_t* const* slot_values,
                              const int* slot_n, int restart_interval_mcus,
                              const int* h_samp, const int* v_samp) {
  uint8_t* p = out;
  *p++ = 0xFF;
  *p++ = 0xD8; /* SOI */
  *p++ = 0xFF;
  *p++ = 0xC3; /* SOF3 */
  put16(&p, 8 + 3 * n_comp);
  *p++ = (uint8_t)prec;
  put16(&p, frame_h);
  put16(&p, frame_w);
  *p++ = (uint8_t)n_comp;
  for (int c = 0; c < n_comp; ++c) {
    *p++ = (uint8_t)(c + 1);
    *p++ = (uint8_t)(((h_samp ? h_samp[c] : 1) << 4) | (v_samp ? v_samp[c] : 1));
    *p++ = 0;
  }
  for (int s = 0; s < n_slots; ++s) {
    *p++ = 0xFF;
    *p++ = 0xC4; /* DHT */
    put16(&p, 2 + 1 + 16 + slot_n[s]);
    *p++ = (uint8_t)s; /* class 0, destination s */
    memcpy(p, slot_counts[s], 16);
    p += 16;
    memcpy(p, slot_values[s], slot_n[s]);
    p += slot_n[s];
  }
  if (restart_interval_mcus > 0) {
    *p++ = 0xFF;
    *p++ = 0xDD; /* DRI */
    put16(&p, 4);
    put16(&p, restart_interval_mcus);
  }
  *p++ = 0xFF;
  *p++ = 0xDA; /* SOS */
  put16(&p, 6 + 2 * n_comp);
  *p++ = (uint8_t)n_comp;
  for (int c = 0; c < n_comp; ++c) {
    *p++ = (uint8_t)(c + 1);
    *p++ = (uint8_t)(comp_table[c] << 4);
  }
  *p++ = 1; /* Ss = predictor 1 */
  *p++ = 0; /* Se */
  *p++ = 0; /* Ah/Al: Pt = 0 */
  return (size_t)(p - out);
}
