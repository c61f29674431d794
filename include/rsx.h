/*
 * rsx.h -- C-ABI of the MI355X-native RAW decompression core ("rsx").
 *
 * This is the drop-in boundary behind rawspeed's decompressor classes.  The
 * reference (darktable-org/rawspeed, paths relative to src/librawspeed/)
 * exposes no plugin/FFI interface for decompressors; the seam is created by
 * forwarding from three concrete C++ methods (plus the DNG tile fan-out) into
 * the entry points below.  Every entry point cites the reference interface it
 * replaces.  Plain C types only: no C++ objects, no torch types, no
 * exceptions cross this boundary.  All pointers are borrowed for the duration
 * of the call.  Entry points are re-entrant per context; one context may be
 * shared by several host threads (calls are serialised on the context).
 *
 * Two families of entry points:
 *   - host-pointer calls (rsx_unpack_u16, rsx_ljpeg_decode, rsx_cr2_decode,
 *     rsx_dng_decompress): exactly what the patched reference methods call;
 *     input is pageable host memory, output is the RawImage's host buffer.
 *     They stage H2D / D2H internally.
 *   - device-resident plans (rsx_*_plan_*): inputs and outputs already live in
 *     HBM, launches go to a caller-supplied hipStream_t.  These are what the
 *     roofline measurement and the batched multi-GPU path use.
 */
#ifndef RSX_H
#define RSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSX_ABI_VERSION 4

/* ------------------------------------------------------------------------ */
/* Status codes.  Kernels cannot throw; the C++ forwarding shim converts a   */
/* non-OK status into ThrowRDE / ThrowIOE.                  */
/* ------------------------------------------------------------------------ */
typedef enum rsx_status {
  RSX_OK = 0,
  /* descriptor rejected by the same checks the reference constructor makes
   * (ThrowRDE in UncompressedDecompressor.cpp:106-169,
   * LJpegDecompressor.cpp:52-152, Cr2DecompressorImpl.h:279-363) */
  RSX_ERR_INVALID_ARG = 1,
  /* not enough input (ThrowIOE: UncompressedDecompressor.cpp:52-74,
   * BitStreamer.h:58-59 "Bit stream size is smaller than MaxProcessBytes") */
  RSX_ERR_IO = 2,
  /* "bad Huffman code" (codes/PrefixCodeLookupDecoder.h:152-155) */
  RSX_ERR_BAD_HUFFMAN_CODE = 3,
  /* restart marker missing / wrong (LJpegDecompressor.cpp:288-297) */
  RSX_ERR_RESTART_MARKER = 4,
  /* "Buffer overflow read in BitStreamer" (bitstreams/BitStreamer.h:125-127) */
  RSX_ERR_INPUT_OVERFLOW = 5,
  /* HIP runtime failure (no reference equivalent) */
  RSX_ERR_DEVICE = 6,
  /* valid for the reference but not implemented by this core yet */
  RSX_ERR_UNSUPPORTED = 7,
  RSX_ERR_NOMEM = 8,
  /* AbstractDngDecompressor: at least one tile failed
   * (AbstractDngDecompressor.cpp:247-251 "Too many errors") */
  RSX_ERR_TILE_ERRORS = 9,
  /* "decoded value out of bounds" (PentaxDecompressor.cpp:170-171) */
  RSX_ERR_VALUE_RANGE = 10
} rsx_status;

/* Bit orders; numeric values equal rawspeed::BitOrder
 * (bitstreams/BitStreams.h:27-35). */
typedef enum rsx_bit_order {
  RSX_ORDER_LSB = 0,
  RSX_ORDER_MSB = 1,
  RSX_ORDER_MSB16 = 2,
  RSX_ORDER_MSB32 = 3,
  RSX_ORDER_JPEG = 4
} rsx_bit_order;

/* ------------------------------------------------------------------------ */
/* The RawImage view (common/RawImage.h:289-296): uint16 samples,            */
/* `pitch_bytes` between rows (never assume roundUp(w*bpp,16):               */
/* RawImage.cpp:85-90), `dim_x` x `dim_y` pixels of `cpp` samples each.      */
/* `data` is a host pointer for the host-pointer calls and a device pointer  */
/* for the plan calls.                                                       */
/* ------------------------------------------------------------------------ */
typedef struct rsx_image {
  void* data;
  uint32_t pitch_bytes;
  int32_t dim_x;
  int32_t dim_y;
  int32_t cpp;
  int32_t is_cfa; /* RawImageData::isCFA; only Cr2 sRaw validation reads it */
} rsx_image;

/* ------------------------------------------------------------------------ */
/* Context: one per (host thread group, device).  Owns a HIP stream, staging */
/* buffers and scratch.                                                      */
/* ------------------------------------------------------------------------ */
typedef struct rsx_ctx rsx_ctx;

int rsx_abi_version(void);
const char* rsx_status_string(int status);
/* Number of visible HIP devices (0 when no GPU / no driver). */
int rsx_device_count(void);
/* Creates a context on HIP device `device`.  Fails with RSX_ERR_DEVICE when
 * there is no usable GPU: there is NO CPU fallback in this library. */
int rsx_ctx_create(int device, rsx_ctx** out_ctx);
void rsx_ctx_destroy(rsx_ctx* ctx);
/* Last error text of this context (never NULL). */
const char* rsx_ctx_last_error(const rsx_ctx* ctx);
/* Number of host-pointer calls (rsx_*_decode / rsx_unpack_* / rsx_dng_decompress_* ...)
 * this context has served: lets an integration check that the batched DNG hunk
 *  really makes one call per image. */
uint64_t rsx_ctx_host_calls(const rsx_ctx* ctx);
/* ... and how many of them ran their one large stream in chunks, the upload and the download
 * under the decode (round 6; LJpegDecoder::decode / Cr2LJpegDecoder::decode of a frame whose plan
 * the calling thread's lane holds from the frame before: LJpegDecoder.cpp:161-164,
 * Cr2LJpegDecoder.cpp:150-153 are the callers): a diagnostic, like the count above. */
uint64_t rsx_ctx_chunked_calls(const rsx_ctx* ctx);

/* Optional: page-locked host memory for the host-pointer calls (ABI 4).
 * Those calls take whatever the caller has -- rawspeed's file `Buffer` and the pixel store
 * of RawImageData::createData() (RawImage.cpp:68-100: `data.resize(pitch * dim.y)` over an
 * aligned allocator) are pageable, and a copy from / to pageable memory goes through the
 * driver's staging at a fraction of the link's rate and keeps its calling thread.  An
 * integration that allocates these two buffers here (or registers them after the fact)
 * gets direct DMA and copies that overlap the kernels; nothing else changes, and memory
 * that was not registered keeps working as before.  INTEGRATION.md 6 shows the hunks.
 *   rsx_host_alloc / rsx_host_free          hipHostMalloc'ed block (64-byte aligned and more)
 *   rsx_host_register / rsx_host_unregister  page-lock an existing allocation in place
 * RSX_ERR_NOMEM when the pages cannot be locked (RLIMIT_MEMLOCK, fragmentation): the
 * caller carries on with pageable memory. */
int rsx_host_alloc(rsx_ctx* ctx, size_t bytes, void** out);
int rsx_host_free(rsx_ctx* ctx, void* p);
int rsx_host_register(rsx_ctx* ctx, void* p, size_t bytes);
int rsx_host_unregister(rsx_ctx* ctx, void* p);

/* ------------------------------------------------------------------------ */
/* 1. UncompressedDecompressor                                               */
/*    replaces UncompressedDecompressor::readUncompressedRaw()               */
/*    (decompressors/UncompressedDecompressor.h:75, .cpp:202-268) for the    */
/*    UINT16 packed-integer paths, i.e. decodePackedInt<BitStreamerXXX>      */
/*    (.cpp:188-200) and the 16-bit-LSB copyPixels fast path (.cpp:255-265). */
/*    Fields mirror the constructor (.h:64-66, .cpp:106-169).                */
/* ------------------------------------------------------------------------ */
typedef struct rsx_unpack_desc {
  int32_t crop_x, crop_y; /* iRectangle2D crop.pos (pixels) */
  int32_t crop_w, crop_h; /* iRectangle2D crop.dim (pixels) */
  int32_t input_pitch_bytes;
  int32_t bits_per_pixel; /* 1..16 for UINT16 images */
  int32_t bit_order;      /* rsx_bit_order, JPEG rejected */
} rsx_unpack_desc;

/* Validation only (the reference constructor): RSX_OK or the error the
 * reference would throw.  Needs no GPU. */
int rsx_unpack_validate(const rsx_unpack_desc* d, const rsx_image* img,
                        size_t in_bytes);

int rsx_unpack_u16(rsx_ctx* ctx, const rsx_unpack_desc* d, const uint8_t* in,
                   size_t in_bytes, const rsx_image* img);

/* 1a. The same method on RawImageType::F32 images (.cpp:212-245):            */
/*    bits_per_pixel 32 -> copyPixels (.cpp:213-222), 16 / 24 with MSB or LSB*/
/*    order -> decodePackedFP<Pump, Binary16 | Binary24> (.cpp:171-186,      */
/*    common/FloatingPoint.h:109-145: exact widening to binary32, subnormals */
/*    renormalised, NaN payloads kept).  Anything else is the reference's    */
/*    "Unsupported floating-point input bitwidth/bit packing".  `img->data`  */
/*    holds 4-byte samples; unlike the integer path, decodePackedFP honours  */
/*    crop_x (as a SAMPLE offset, .cpp:181) -- replicated.                   */
int rsx_unpack_f32_validate(const rsx_unpack_desc* d, const rsx_image* img,
                            size_t in_bytes);
int rsx_unpack_f32(rsx_ctx* ctx, const rsx_unpack_desc* d, const uint8_t* in,
                   size_t in_bytes, const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 1b. The fixed-layout entry points of the same class                       */
/*    UncompressedDecompressor::decode8BitRaw<true>()        (.cpp:270-291)  */
/*    UncompressedDecompressor::decode12BitRawWithControl<e>() (.cpp:296-349)*/
/*    UncompressedDecompressor::decode12BitRawUnpackedLeftAligned<e>()       */
/*                                                           (.cpp:356-378)  */
/*    They use only size = crop.dim of the constructor and write from pixel  */
/*    (0,0) of the image, ignoring crop.pos -- replicated.                   */
/*    decode8BitRaw<false>() stores setWithLookUp(byte) instead              */
/*    (common/RawImage.h:335-353) with a random state that starts at 0 and   */
/*    therefore stays 0: the result is a pure function of the byte, passed   */
/*    as `lut` (for a dithering table: tables[2 * v], else tables[v]).       */
/* ------------------------------------------------------------------------ */
typedef enum rsx_unpack_variant {
  RSX_UNPACK_8BIT_RAW = 0,
  RSX_UNPACK_12BIT_WITH_CONTROL = 1,
  RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED = 2,
  RSX_UNPACK_8BIT_LOOKUP = 3 /* decode8BitRaw<false> */
} rsx_unpack_variant;

typedef struct rsx_unpack_variant_desc {
  int32_t variant;    /* rsx_unpack_variant */
  int32_t big_endian; /* template parameter Endianness e (ignored for 8-bit) */
  int32_t w, h;       /* size.x, size.y */
  uint16_t lut[256];  /* RSX_UNPACK_8BIT_LOOKUP only */
} rsx_unpack_variant_desc;

int rsx_unpack_variant_validate(const rsx_unpack_variant_desc* d,
                                const rsx_image* img, size_t in_bytes);
int rsx_unpack_variant_u16(rsx_ctx* ctx, const rsx_unpack_variant_desc* d,
                           const uint8_t* in, size_t in_bytes,
                           const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* Huffman table exactly as the DHT payload the host already parsed          */
/* (codes/HuffmanCode.h:99-166): 16 counts + the code values (= SSSS         */
/* difference categories, each <= 16 in full-decode mode,                    */
/* codes/AbstractPrefixCodeTranscoder.h:71-84).  The canonical code is       */
/* re-derived on our side; the reference's LUT memory is never read.         */
/* ------------------------------------------------------------------------ */
#define RSX_MAX_CODE_VALUES 162 /* codes/AbstractPrefixCode.h BaselineCodeTag */
typedef struct rsx_huff_table {
  uint8_t n_codes_per_length[16]; /* index 0 = code length 1 */
  uint8_t code_values[RSX_MAX_CODE_VALUES];
  uint8_t n_code_values;
  uint8_t fix_dng_bug16; /* AbstractPrefixCodeDecoder.h:58-62 */
} rsx_huff_table;

#define RSX_MAX_COMPONENTS 4

/* ------------------------------------------------------------------------ */
/* 2. LJpegDecompressor                                                      */
/*    replaces LJpegDecompressor::decode() (LJpegDecompressor.h:94,          */
/*    .cpp:341-370 -> decodeN .cpp:254-339 -> decodeRowN .cpp:184-251).      */
/*    Fields mirror the constructor (.h:89-93, .cpp:52-152).                 */
/* ------------------------------------------------------------------------ */
typedef struct rsx_ljpeg_desc {
  int32_t tile_x, tile_y, tile_w, tile_h; /* imgFrame, pixels */
  int32_t mcu_w, mcu_h;                   /* Frame::mcu */
  int32_t frame_w, frame_h;               /* Frame::dim, in MCUs */
  int32_t n_comp;                         /* rec.size() == mcu_w*mcu_h */
  int32_t rows_per_restart_interval;      /* numLJpegRowsPerRestartInterval */
  uint16_t init_pred[RSX_MAX_COMPONENTS]; /* PerComponentRecipe::initPred */
  uint8_t table_index[RSX_MAX_COMPONENTS]; /* PerComponentRecipe::ht -> tables[] */
  int32_t n_tables;
  rsx_huff_table tables[RSX_MAX_COMPONENTS];
} rsx_ljpeg_desc;

int rsx_ljpeg_validate(const rsx_ljpeg_desc* d, const rsx_image* img,
                       size_t in_bytes);

/* `in` = entropy-coded data from just after the SOS header to the end of the
 * tile buffer (what LJpegDecoder::decodeScan passes, LJpegDecoder.cpp:161-164).
 * `consumed` = ByteStream::size_type return value of decode() (closed form:
 * SURVEY.md A.6). */
int rsx_ljpeg_decode(rsx_ctx* ctx, const rsx_ljpeg_desc* d, const uint8_t* in,
                     size_t in_bytes, const rsx_image* img,
                     uint32_t* consumed);

/* ------------------------------------------------------------------------ */
/* 3. Cr2Decompressor<PrefixCodeDecoder<>>                                   */
/*    replaces Cr2Decompressor::decompress() (Cr2Decompressor.h:174,         */
/*    Cr2DecompressorImpl.h:471-485 -> decompressN_X_Y :396-468).            */
/*    Fields mirror the constructor (Cr2Decompressor.h:168-172,              */
/*    Cr2DecompressorImpl.h:279-363).                                        */
/* ------------------------------------------------------------------------ */
typedef struct rsx_cr2_desc {
  int32_t n_comp, x_s_f, y_s_f; /* format tuple */
  int32_t frame_w, frame_h;     /* iPoint2D frame (as passed, before /X_S_F) */
  int32_t num_slices, slice_width, last_slice_width; /* Cr2SliceWidths */
  uint16_t init_pred[RSX_MAX_COMPONENTS];
  uint8_t table_index[RSX_MAX_COMPONENTS];
  int32_t n_tables;
  rsx_huff_table tables[RSX_MAX_COMPONENTS];
} rsx_cr2_desc;

int rsx_cr2_validate(const rsx_cr2_desc* d, const rsx_image* img,
                     size_t in_bytes);

int rsx_cr2_decode(rsx_ctx* ctx, const rsx_cr2_desc* d, const uint8_t* in,
                   size_t in_bytes, const rsx_image* img, uint32_t* consumed);

/* ------------------------------------------------------------------------ */
/* 3a. Cr2sRawInterpolator                                                   */
/*    replaces Cr2sRawInterpolator::interpolate(version)                     */
/*    (interpolators/Cr2sRawInterpolator.h:49, .cpp:510-542 ->               */
/*    interpolate_422<v> :95-186 / interpolate_420<v> :188-460,              */
/*    YUV_TO_RGB<v> :470-506): the step Cr2Decoder runs right after the sRaw */
/*    decompress (Cr2Decoder.cpp:585-625).  `in` is the decoded subsampled   */
/*    image (cpp 1: groups of Y Y Cb Cr or Y Y Y Y Cb Cr), `out` the         */
/*    interpolated one (cpp 3, 2 * groups pixels wide, subsampling_y * rows  */
/*    high).  All arithmetic is the reference's int arithmetic.              */
/* ------------------------------------------------------------------------ */
typedef struct rsx_sraw_desc {
  int32_t version;        /* 0, 1, 2 (0 only with subsampling_y == 1) */
  int32_t subsampling_y;  /* 1: 4:2:2 (groups of 4), 2: 4:2:0 (groups of 6); x is always 2 */
  int32_t sraw_coeffs[3];
  int32_t hue;
} rsx_sraw_desc;

int rsx_sraw_validate(const rsx_sraw_desc* d, const rsx_image* in, const rsx_image* out);
int rsx_sraw_interpolate(rsx_ctx* ctx, const rsx_sraw_desc* d, const rsx_image* in,
                         const rsx_image* out);

/* ------------------------------------------------------------------------ */
/* 3b. NikonDecompressor                                                     */
/*    replaces NikonDecompressor::decompress(input, uncorrectedRawValues)    */
/*    (decompressors/NikonDecompressor.h:57, .cpp:541-560 ->                 */
/*    decompress<Huffman>(bits, start_y, end_y) :515-539).  The constructor  */
/*    (.cpp:473-513: metadata parsing, createCurve :381-445) stays on the    */
/*    host; its results are the fields below.                                */
/*      - bit stream: BitStreamerMSB over `in` (no byte stuffing, no markers)*/
/*      - tables[0] = nikon_tree[huffSelect] decoded by PrefixCodeDecoder<>  */
/*        (full decode, no DNG bug); rows >= split (if split != 0) use       */
/*        tables[1] = nikon_tree[huffSelect + 1] with the "lossy after       */
/*        split" semantics of NikonLASDecompressor::decodeDifference         */
/*        (.cpp:331-376): code value v -> len = v & 15, shl = v >> 4,        */
/*        len - shl raw bits, diff = ((bits << 1) + 1) << shl >> 1, sign     */
/*        extension on bit len-1; v == 16 -> -32768                          */
/*      - predictor: int accumulators pred[col & 1] seeded from              */
/*        pUp[row & 1][col & 1], which the first two columns update          */
/*      - output: clampBits(pred, 15), then RawImageData::setWithLookUp      */
/*        (common/RawImage.h:335-353): as is when uncorrected_raw_values,    */
/*        else through the dithering TableLookUp built from `curve`          */
/*        (common/TableLookUp.cpp:50-84) with the serial random state seeded */
/*        from bits.peekBits(24) (.cpp:549)                                  */
/* ------------------------------------------------------------------------ */
typedef struct rsx_nikon_desc {
  int32_t bits_ps; /* 12 or 14 (.cpp:485-491) */
  int32_t split;   /* 0 = no split (already clamped against dim.y, .cpp:511-512) */
  int32_t p_up[2][2]; /* pUp[row & 1][col & 1] (.cpp:506-509) */
  int32_t uncorrected_raw_values;
  int32_t curve_size;    /* entries of `curve` (1 .. 65536) */
  const uint16_t* curve; /* host pointer; copied during the call */
  rsx_huff_table tables[2];
} rsx_nikon_desc;

int rsx_nikon_validate(const rsx_nikon_desc* d, const rsx_image* img);

int rsx_nikon_decompress(rsx_ctx* ctx, const rsx_nikon_desc* d, const uint8_t* in,
                         size_t in_bytes, const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 3c. PentaxDecompressor                                                    */
/*    replaces PentaxDecompressor::decompress(ByteStream data)               */
/*    (decompressors/PentaxDecompressor.h, .cpp:152-176): BitStreamerMSB,    */
/*    one PrefixCodeDecoder<> (the legacy tree or the one the constructor    */
/*    derives from the makernote, .cpp:69-150 -- stays on the host),         */
/*    pred[col & 1] += diff with both predictors starting from the pixels    */
/*    two rows up (0 for the first two rows); a value outside [0, 65535]     */
/*    is RSX_ERR_VALUE_RANGE.                                                */
/* ------------------------------------------------------------------------ */
typedef struct rsx_pentax_desc {
  rsx_huff_table table;
} rsx_pentax_desc;

int rsx_pentax_validate(const rsx_pentax_desc* d, const rsx_image* img);
int rsx_pentax_decompress(rsx_ctx* ctx, const rsx_pentax_desc* d, const uint8_t* in,
                          size_t in_bytes, const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 3d. SamsungV1Decompressor                                                 */
/*    replaces SamsungV1Decompressor::decompress()                           */
/*    (decompressors/SamsungV1Decompressor.cpp:81-140): BitStreamerMSB, a    */
/*    fixed prefix code given as (encLen, diffLen) pairs that fill a 10-bit  */
/*    table in order (.cpp:88-117) -- NOT a canonical JPEG code, so it is    */
/*    handed over in that form --, samsungDiff (.cpp:63-79: fill(23),        */
/*    diffLen raw bits, sign extension), and the two-rows-up predictor of    */
/*    PentaxDecompressor with values limited to `bits` bits (.cpp:129-136).  */
/* ------------------------------------------------------------------------ */
#define RSX_SAMSUNG_V1_MAX_ENTRIES 32
typedef struct rsx_samsung_v1_desc {
  int32_t bits;      /* the constructor accepts only 12 (.cpp:53-54) */
  int32_t n_entries; /* 14 in the reference */
  uint8_t enc_len[RSX_SAMSUNG_V1_MAX_ENTRIES];  /* tab[i][0], 1..10 */
  uint8_t diff_len[RSX_SAMSUNG_V1_MAX_ENTRIES]; /* tab[i][1], 0..13 */
} rsx_samsung_v1_desc;

int rsx_samsung_v1_validate(const rsx_samsung_v1_desc* d, const rsx_image* img);
int rsx_samsung_v1_decompress(rsx_ctx* ctx, const rsx_samsung_v1_desc* d,
                              const uint8_t* in, size_t in_bytes, const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 3g. SamsungV2Decompressor                                                 */
/*    replaces SamsungV2Decompressor::decompress()                           */
/*    (decompressors/SamsungV2Decompressor.h:74, .cpp:340-343: decompressRow */
/*    for every row, .cpp:312-338).  The descriptor is what the constructor  */
/*    (.cpp:85-141) reads from the 16-byte header -- bitDepth, width,        */
/*    height, optflags, initVal -- and `in` is its member `data`: the bytes   */
/*    behind that header.  Every row is a BitStreamerMSB32 of its own that   */
/*    starts at the next 16-byte boundary of `data` (.cpp:314-318); a row is */
/*    width / 16 blocks (processBlock .cpp:312-327: prepareBaselineValues    */
/*    .cpp:152-230, decodeDiffLengths .cpp:232-277, decodeDifferences        */
/*    .cpp:279-311), pixels are clampBits(baseline + difference, bitDepth).  */
/*    rsx_samsung_v2_validate = the constructor's checks (.cpp:88-100,       */
/*    :123-125, :134-139).  Statuses: what the reference throws, in its      */
/*    order -- RSX_ERR_INVALID_ARG for its ThrowRDEs (.cpp:172-173, :191-    */
/*    192, :212-219, :258-259, :271-272), RSX_ERR_IO / RSX_ERR_INPUT_        */
/*    OVERFLOW for the stream running out; the image is unspecified then.    */
/* ------------------------------------------------------------------------ */
typedef struct rsx_samsung_v2_desc {
  int32_t bit_depth;  /* bitDepth: 12 or 14 */
  int32_t width;      /* multiple of 16, <= 6496 */
  int32_t height;     /* <= 4336 */
  uint32_t optflags;  /* OptFlags (.cpp:46-54): 1 SKIP, 2 MV, 4 QP */
  uint32_t init_val;  /* initVal (14 bits) */
} rsx_samsung_v2_desc;

int rsx_samsung_v2_validate(const rsx_samsung_v2_desc* d, const rsx_image* img);
int rsx_samsung_v2_decompress(rsx_ctx* ctx, const rsx_samsung_v2_desc* d, const uint8_t* in,
                              size_t in_bytes, const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 3e. HasselbladDecompressor                                                */
/*    replaces HasselbladDecompressor::decompress()                          */
/*    (decompressors/HasselbladDecompressor.h:53, .cpp:71-100): BitStreamer- */
/*    MSB32 (little-endian 32-bit words, MSB first, no stuffing), pixels     */
/*    coded two at a time as [len1 code][len2 code][len1 bits][len2 bits]    */
/*    with one PrefixCodeDecoder<> used for its code VALUES only             */
/*    (decodeCodeValue), getBits (.cpp:60-69: JPEG sign extension, all-ones  */
/*    16-bit field = -32768), both predictors restart from initPred on every */
/*    row, results stored mod 2^16.  Returns getStreamPosition().            */
/* ------------------------------------------------------------------------ */
typedef struct rsx_hasselblad_desc {
  rsx_huff_table table; /* rec.ht (values = difference lengths 0..16) */
  uint16_t init_pred;   /* rec.initPred */
} rsx_hasselblad_desc;

int rsx_hasselblad_validate(const rsx_hasselblad_desc* d, const rsx_image* img);
int rsx_hasselblad_decompress(rsx_ctx* ctx, const rsx_hasselblad_desc* d, const uint8_t* in,
                              size_t in_bytes, const rsx_image* img, uint32_t* consumed);

/* ------------------------------------------------------------------------ */
/* 3f. SonyArw1Decompressor                                                  */
/*    replaces SonyArw1Decompressor::decompress(ByteStream)                  */
/*    (decompressors/SonyArw1Decompressor.h:44, .cpp:59-93): BitStreamerMSB, */
/*    a fixed prefix code for the difference length (2 bits -> 4 - x; "011"  */
/*    = 0; "00" + k zeros + "1" = 4 + k, capped at 17, .cpp:76-83), JPEG     */
/*    sign extension of the difference bits, ONE predictor running through   */
/*    the whole image in decode order: columns right to left, in each column */
/*    the even rows top to bottom and then the odd rows (.cpp:68-74).  A     */
/*    value outside 0..4095 is an error (isIntN(pred, 12), .cpp:88-89).  The */
/*    constructor's checks (cpp 1, U16, w <= 4600, h <= 3072, h even,        */
/*    .cpp:39-51) are rsx_sony_arw1_validate.  No parameters besides the     */
/*    image: the code is fixed.                                              */
/* ------------------------------------------------------------------------ */
int rsx_sony_arw1_validate(const rsx_image* img);
int rsx_sony_arw1_decompress(rsx_ctx* ctx, const uint8_t* in, size_t in_bytes,
                             const rsx_image* img);

/* ------------------------------------------------------------------------ */
/* 4. AbstractDngDecompressor tile fan-out                                   */
/*    replaces AbstractDngDecompressor::decompress()                         */
/*    (AbstractDngDecompressor.h:141, .cpp:240-252) for compression 1        */
/*    (decompressThread<1> .cpp:54-110: per-tile UncompressedDecompressor)   */
/*    and compression 7 (decompressThread<7> .cpp:112-131: per-tile          */
/*    LJpegDecoder; the host keeps parsing each tile's JPEG headers and      */
/*    hands us one rsx_ljpeg_desc per tile).  All tiles are decoded by ONE   */
/*    batched launch sequence instead of one OpenMP thread per tile.         */
/*    Per-tile failures are reported in tile_status[] (the reference appends */
/*    them to the ErrorLog, .cpp:122-129); the call returns                  */
/*    RSX_ERR_TILE_ERRORS if any tile failed (.cpp:247-251).                 */
/* ------------------------------------------------------------------------ */
typedef struct rsx_dng_ljpeg_tile {
  rsx_ljpeg_desc desc;
  const uint8_t* in; /* entropy-coded data of this tile */
  size_t in_bytes;
} rsx_dng_ljpeg_tile;

int rsx_dng_decompress_ljpeg(rsx_ctx* ctx, int n_tiles,
                             const rsx_dng_ljpeg_tile* tiles,
                             const rsx_image* img, int32_t* tile_status,
                             uint32_t* tile_consumed);

typedef struct rsx_dng_unpack_tile {
  rsx_unpack_desc desc;
  const uint8_t* in;
  size_t in_bytes;
} rsx_dng_unpack_tile;

int rsx_dng_decompress_uncompressed(rsx_ctx* ctx, int n_tiles,
                                    const rsx_dng_unpack_tile* tiles,
                                    const rsx_image* img,
                                    int32_t* tile_status);

/* ------------------------------------------------------------------------ */
/* Device-resident plans (inputs/outputs already in HBM).                    */
/*                                                                           */
/* A plan is "validate + size scratch + upload tables once, launch many".   */
/* `in_dev`/`out_dev` are device base pointers; each job addresses           */
/* [in_offset, in_offset+in_bytes) of the input and an image view starting  */
/* `img_offset` bytes into the output.  `stream` is a hipStream_t (may be    */
/* NULL = the context's stream).  run() only enqueues; results() blocks on   */
/* the stream, then reports per-job status / consumed byte counts.           */
/* The context's stream is hipStreamNonBlocking: with stream == NULL a run   */
/* is queued behind the work the caller has put on the NULL stream so far    */
/* (an event wait); a caller who passes a stream orders the run himself --   */
/* in_dev must be final and out_dev free on THAT stream.                     */
/* ------------------------------------------------------------------------ */
typedef struct rsx_plan rsx_plan;

typedef struct rsx_unpack_job {
  rsx_unpack_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_unpack_job;

typedef struct rsx_unpack_variant_job {
  rsx_unpack_variant_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_unpack_variant_job;

typedef struct rsx_ljpeg_job {
  rsx_ljpeg_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_ljpeg_job;

typedef struct rsx_cr2_job {
  rsx_cr2_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_cr2_job;

typedef struct rsx_nikon_job {
  rsx_nikon_desc desc; /* desc.curve: host pointer, copied at plan creation */
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_nikon_job;

typedef struct rsx_pentax_job {
  rsx_pentax_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_pentax_job;

/* in_offset / img_offset address the subsampled input image and the interpolated
 * output image inside the plan's input / output buffers */
typedef struct rsx_sraw_job {
  rsx_sraw_desc desc;
  uint64_t in_offset;
  uint64_t img_offset;
  rsx_image in;  /* .data ignored */
  rsx_image img; /* .data ignored */
} rsx_sraw_job;

typedef struct rsx_hasselblad_job {
  rsx_hasselblad_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_hasselblad_job;

typedef struct rsx_samsung_v1_job {
  rsx_samsung_v1_desc desc;
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_samsung_v1_job;

typedef struct rsx_samsung_v2_job {
  rsx_samsung_v2_desc desc;
  uint64_t in_offset; /* multiple of 16: the rows start at 16-byte boundaries of the data */
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_samsung_v2_job;

typedef struct rsx_sony_arw1_job {
  uint64_t in_offset;
  uint64_t in_bytes;
  uint64_t img_offset;
  rsx_image img; /* .data ignored */
} rsx_sony_arw1_job;

int rsx_unpack_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_unpack_job* jobs,
                           rsx_plan** out_plan);
/* F32 images: same job structure, img describes 4-byte samples */
int rsx_unpack_f32_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_unpack_job* jobs,
                               rsx_plan** out_plan);
int rsx_unpack_variant_plan_create(rsx_ctx* ctx, int n_jobs,
                                   const rsx_unpack_variant_job* jobs,
                                   rsx_plan** out_plan);
int rsx_ljpeg_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_ljpeg_job* jobs,
                          rsx_plan** out_plan);
int rsx_cr2_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_cr2_job* jobs,
                        rsx_plan** out_plan);
/* (jobs with split != 0 cost one host round trip per run: the second table
 * starts at a bit position only known once the first part is decoded) */
int rsx_nikon_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_nikon_job* jobs,
                          rsx_plan** out_plan);
int rsx_pentax_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_pentax_job* jobs,
                           rsx_plan** out_plan);
int rsx_samsung_v1_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_samsung_v1_job* jobs,
                               rsx_plan** out_plan);
int rsx_samsung_v2_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_samsung_v2_job* jobs,
                               rsx_plan** out_plan);
int rsx_sraw_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_sraw_job* jobs,
                         rsx_plan** out_plan);
int rsx_hasselblad_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_hasselblad_job* jobs,
                               rsx_plan** out_plan);
int rsx_sony_arw1_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_sony_arw1_job* jobs,
                              rsx_plan** out_plan);
/* Enqueue one pass of the plan on `stream`. */
int rsx_plan_run(rsx_plan* plan, const void* in_dev, void* out_dev,
                 void* stream);
/* Wait for the last run and fetch per-job results.  `job_status` and
 * `job_consumed` may be NULL.  Returns RSX_OK iff every job is RSX_OK. */
int rsx_plan_results(rsx_plan* plan, int32_t* job_status,
                     uint32_t* job_consumed);
/* Name of the dominant kernel of this plan and the average duration (ms) of
 * its launches since the previous call, measured with hipEvents recorded on
 * the stream the kernel is launched on.  Timing is off by default; enable
 * with rsx_plan_set_timing(plan, 1).  Returns RSX_OK, or RSX_ERR_INVALID_ARG
 * if timing is disabled / no launches happened. */
int rsx_plan_set_timing(rsx_plan* plan, int enable);
int rsx_plan_kernel_time(rsx_plan* plan, const char** kernel_name,
                         double* avg_ms, int* n_launches);
/* LJPEG-family plans run a sequence of kernels.  With timing enabled an event is
 * recorded after every launch of a run (on the launch stream), and
 * rsx_plan_kernel_time() names the kernel with the largest total.  This call
 * returns the whole table -- up to `cap` kernel names with their average
 * duration per run (ms) over the timed runs since timing was enabled -- so that
 * the dominant kernel can be checked rather than believed.  *n_kernels = entries
 * the plan has (may exceed cap), *n_runs = runs averaged over. */
int rsx_plan_kernel_table(rsx_plan* plan, int cap, const char** names, double* avg_ms,
                          int* n_kernels, int* n_runs);
void rsx_plan_destroy(rsx_plan* plan);

/* Measurement aid, not part of the decode path: runs a plain streaming kernel
 * (16-byte loads of in_bytes, 16-byte stores of out_bytes, device pointers) `reps`
 * times on `stream` and returns the average duration in ms, measured with
 * hipEvents on that stream.  bench.py uses it to report the copy ceiling of the
 * device for the read:write mix of an unpack launch next to the vendor peak
 * ("also report a measured device-copy ceiling"). */
int rsx_probe_stream_copy(rsx_ctx* ctx, const void* in_dev, size_t in_bytes,
                          void* out_dev, size_t out_bytes, void* stream, int reps,
                          double* avg_ms);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* RSX_H */
