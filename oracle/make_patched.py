#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates the *patched* copies of the three reference
translation units that INTEGRATION.md describes, so that the drop-in boundary
can be exercised for real: reference host code (JPEG header parsing, DNG tile
fan-out, RawImage) -> forwarding hunk -> librsx.so -> MI355X.

Inputs are read from /root/reference; outputs are written ONLY under
oracle/_ref/patched/ (git-ignored).  Nothing of the reference is stored in the
repository: this script holds our hunks and the one-line anchors they attach to.
"""
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
S = os.path.join(REF, "src", "librawspeed")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "patched")

# Every hunk has the same shape: forward to the MI355X core and return when it
# succeeded; on ANY other status -- no device, unsupported shape, no memory, a damaged
# stream -- fall through to the method's original body, which does the work on the CPU
# or throws exactly what the reference throws (message, partial image, ErrorLog).  The
# C-ABI copies back only what a successful decode produced, so the image is untouched
# when the original body takes over.
HUNK_UNPACK = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_unpack_desc d{};
    d.crop_x = offset.x;
    d.crop_y = offset.y;
    d.crop_w = size.x;
    d.crop_h = size.y;
    d.input_pitch_bytes = inputPitchBytes;
    d.bits_per_pixel = bitPerPixel;
    d.bit_order = static_cast<int32_t>(order);
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    const bool f32 = mRaw->getDataType() == RawImageType::F32;
    // a tile of a DNG whose tiles are decoded by one batched call (INTEGRATION.md 4)
    if (const auto b = rsx_shim::DngBatch::find(in.begin()); b.batch && !f32) {
      if (!b.batch->replaying) {
        b.slot->kind = 2;
        b.slot->up.desc = d;
        b.slot->up.in = in.begin();
        b.slot->up.in_bytes = in.getSize();
        return;
      }
      if (b.slot->status == RSX_OK)
        return;
    } else if (rsx_shim::done((f32 ? rsx_unpack_f32(rsx, &d, in.begin(), in.getSize(), &img)
                    : rsx_unpack_u16(rsx, &d, in.begin(), in.getSize(), &img)))) {
      return;
    }
  }
'''

HUNK_VARIANT = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1b) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_unpack_variant_desc d{};
    d.variant = %(variant)s;
    d.big_endian = %(big)s;
    d.w = size.x;
    d.h = size.y;
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    if (rsx_shim::done(rsx_unpack_variant_u16(rsx, &d, in.begin(), in.getSize(), &img))) {
      input.skipBytes(input.getRemainSize());
      return;
    }
  }
'''

HUNK_8BIT = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1b) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_unpack_variant_desc d{};
    d.variant = uncorrectedRawValues ? RSX_UNPACK_8BIT_RAW : RSX_UNPACK_8BIT_LOOKUP;
    d.w = size.x;
    d.h = size.y;
    if constexpr (!uncorrectedRawValues) {
      // the loop below starts with random = 0, which setWithLookUp maps to 0 again:
      // the stored value is a pure function of the byte
      for (int v = 0; v < 256; ++v) {
        uint16_t px = 0;
        uint32_t rnd = 0;
        mRaw->setWithLookUp(implicit_cast<uint16_t>(v), reinterpret_cast<std::byte*>(&px), &rnd);
        d.lut[v] = px;
      }
    }
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    if (rsx_shim::done(rsx_unpack_variant_u16(rsx, &d, in.begin(), in.getSize(), &img))) {
      input.skipBytes(input.getRemainSize());
      return;
    }
  }
'''
HUNK_CONTROL = HUNK_VARIANT % dict(variant="RSX_UNPACK_12BIT_WITH_CONTROL",
                                   big="e == Endianness::big")
HUNK_LEFT = HUNK_VARIANT % dict(variant="RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED",
                                big="e == Endianness::big")

HUNK_LJPEG = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 2) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_ljpeg_desc d{};
    d.tile_x = imgFrame.pos.x;
    d.tile_y = imgFrame.pos.y;
    d.tile_w = imgFrame.dim.x;
    d.tile_h = imgFrame.dim.y;
    d.mcu_w = frame.mcu.x;
    d.mcu_h = frame.mcu.y;
    d.frame_w = frame.dim.x;
    d.frame_h = frame.dim.y;
    d.n_comp = implicit_cast<int32_t>(rec.size());
    d.rows_per_restart_interval = numLJpegRowsPerRestartInterval;
    rsx_shim::recipes(rec, &d);
    // a tile of a DNG whose tiles are decoded by one batched call (INTEGRATION.md 4):
    // record it; LJpegDecoder's marker walk goes on from the end of the scan
    if (const auto b = rsx_shim::DngBatch::find(input.begin()); b.batch) {
      if (!b.batch->replaying) {
        b.slot->kind = 1;
        b.slot->lj.desc = d;
        b.slot->lj.in = input.begin();
        b.slot->lj.in_bytes = implicit_cast<size_t>(input.size());
        b.slot->consumed = rsx_shim::DngBatch::endOfScan(
            input.begin(), implicit_cast<size_t>(input.size()),
            /*full_height=*/d.tile_h >= d.frame_h * d.mcu_h);
        return b.slot->consumed;
      }
      if (b.slot->status == RSX_OK)
        return b.slot->consumed;
    } else {
      const rsx_image img = rsx_shim::view(mRaw);
      uint32_t consumed = 0;
      if (rsx_shim::done(rsx_ljpeg_decode(rsx, &d, input.begin(), implicit_cast<size_t>(input.size()), &img,
                           &consumed)))
        return consumed;
    }
  }
'''

HUNK_CR2 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    // the constructor divided frame / slice widths by the sampling factors
    // (Cr2DecompressorImpl.h:305-341); the C-ABI takes them as in the file
    rsx_cr2_desc d{};
    d.n_comp = std::get<0>(format);
    d.x_s_f = std::get<1>(format);
    d.y_s_f = std::get<2>(format);
    d.frame_w = frame.x * d.x_s_f;
    d.frame_h = frame.y * d.y_s_f;
    d.num_slices = slicing.numSlices;
    d.slice_width = slicing.sliceWidth * d.n_comp * d.x_s_f;
    d.last_slice_width = slicing.lastSliceWidth * d.n_comp * d.x_s_f;
    rsx_shim::recipes(rec, &d);
    const rsx_image img = rsx_shim::view(mRaw);
    uint32_t consumed = 0;
    if (rsx_shim::done(rsx_cr2_decode(rsx, &d, input.begin(), implicit_cast<size_t>(input.size()), &img,
                       &consumed)))
      return consumed;
  }
'''

HUNK_NIKON = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3b) ----
  // (after the method's own RawImageCurveGuard: the table state it sets up and leaves
  // behind is the reference's, whatever path decodes)
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_nikon_desc d{};
    d.bits_ps = implicit_cast<int32_t>(bitsPS);
    d.split = implicit_cast<int32_t>(split);
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 2; ++c)
        d.p_up[r][c] = pUp[r][c];
    d.uncorrected_raw_values = uncorrectedRawValues;
    d.curve = curve.data();
    d.curve_size = implicit_cast<int32_t>(curve.size());
    for (int t = 0; t < (split ? 2 : 1); ++t) {
      const auto& tree = nikon_tree[huffSelect + t];
      int n = 0;
      for (int i = 0; i < 16; ++i) {
        d.tables[t].n_codes_per_length[i] = tree[0][i];
        n += tree[0][i];
      }
      for (int i = 0; i < n; ++i)
        d.tables[t].code_values[i] = tree[1][i];
      d.tables[t].n_code_values = implicit_cast<uint8_t>(n);
    }
    const rsx_image img = rsx_shim::view(mRaw);
    if (rsx_shim::done(rsx_nikon_decompress(rsx, &d, input.begin(), implicit_cast<size_t>(input.size()),
                             &img)))
      return;
  }
'''

HUNK_PENTAX = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3c) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_pentax_desc d{};
    d.table = rsx_shim::table(ht);
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = data.peekRemainingBuffer();
    if (rsx_shim::done(rsx_pentax_decompress(rsx, &d, in.begin(), in.getSize(), &img)))
      return;
  }
'''

HUNK_SAMSUNG_V1 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3d) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    // the encoding table of this method (same pairs as `tab` below)
    static const std::array<std::array<uint8_t, 2>, 14> rsx_tab = {{{3, 4}, {3, 7}, {2, 6}, {2, 5},
        {4, 3}, {6, 0}, {7, 9}, {8, 10}, {9, 11}, {10, 12}, {10, 13}, {5, 1}, {4, 8}, {4, 2}}};
    rsx_samsung_v1_desc d{};
    d.bits = bits;
    d.n_entries = 14;
    for (int i = 0; i < 14; ++i) {
      d.enc_len[i] = rsx_tab[i][0];
      d.diff_len[i] = rsx_tab[i][1];
    }
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = bs.peekRemainingBuffer();
    if (rsx_shim::done(rsx_samsung_v1_decompress(rsx, &d, in.begin(), in.getSize(), &img)))
      return;
  }
'''

HUNK_SAMSUNG_V2 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3g) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_samsung_v2_desc d{};
    d.bit_depth = implicit_cast<int32_t>(bitDepth);
    d.width = width;
    d.height = height;
    d.optflags = static_cast<uint32_t>(optflags);
    d.init_val = initVal;
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = data.peekRemainingBuffer();
    if (rsx_shim::done(rsx_samsung_v2_decompress(rsx, &d, in.begin(), in.getSize(), &img)))
      return;
  }
'''

HUNK_SRAW = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3a) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rsx_sraw_desc d{};
    d.version = version;
    d.subsampling_y = mRaw->metadata.subsampling.y;
    if (mRaw->metadata.subsampling.x != 2)
      ThrowRDE("Unknown subsampling: (%i; %i)", mRaw->metadata.subsampling.x,
               mRaw->metadata.subsampling.y);
    for (int c = 0; c < 3; ++c)
      d.sraw_coeffs[c] = sraw_coeffs[c];
    d.hue = hue;
    rsx_image in{};
    in.data = const_cast<uint16_t*>(&input(0, 0));
    in.pitch_bytes = implicit_cast<uint32_t>(input.pitch() * sizeof(uint16_t));
    in.dim_x = input.width();
    in.dim_y = input.height();
    in.cpp = 1;
    const rsx_image out = rsx_shim::view(mRaw);
    if (rsx_shim::done(rsx_sraw_interpolate(rsx, &d, &in, &out)))
      return;
  }
'''

HUNK_SONY_ARW1 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3f) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    if (rsx_shim::done(rsx_sony_arw1_decompress(rsx, in.begin(), in.getSize(), &img)))
      return;
  }
'''

HUNK_HASSELBLAD = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3e) ----
  if (rsx_ctx* rsx = rsx_shim::context()) {
    rec.ht.verifyCodeValuesAsDiffLengths();
    rsx_hasselblad_desc d{};
    d.table = rsx_shim::table(rec.ht);
    d.init_pred = rec.initPred;
    const rsx_image img = rsx_shim::view(mRaw);
    uint32_t consumed = 0;
    if (rsx_shim::done(rsx_hasselblad_decompress(rsx, &d, input.begin(), implicit_cast<size_t>(input.size()),
                                  &img, &consumed)))
      return consumed;
  }
'''

HUNK_DNG = r'''
  // ---- rsx: one batched call for all tiles (INTEGRATION.md 4) ----
  // The fan-out below runs as it is, but while `batch` is registered for the tiles'
  // input ranges the per-tile hunks of UncompressedDecompressor / LJpegDecompressor
  // record their work instead of doing it; run() then makes ONE
  // rsx_dng_decompress_* call.  Tiles it did not finish get a second pass of the same
  // fan-out on the CPU (the finished ones return at once).
  bool rsxDone = false;
  if ((compression == 1 || compression == 7) && rsx_shim::context() != nullptr) {
    rsx_shim::DngBatch batch;
    for (const auto& e : slices) {
      const Buffer b = e.bs.peekRemainingBuffer();
      batch.add(b.begin(), b.getSize());
    }
    batch.registerTiles();
#ifdef HAVE_OPENMP
#pragma omp parallel default(none) num_threads(                                \
        rawspeed_get_number_of_processor_cores()) if (slices.size() > 1)
#endif
    decompressThread();
    if (!batch.run(rsx_shim::view(mRaw))) {
      batch.replaying = true;
#ifdef HAVE_OPENMP
#pragma omp parallel default(none) num_threads(                                \
        rawspeed_get_number_of_processor_cores()) if (slices.size() > 1)
#endif
      decompressThread();
    }
    rsxDone = true;
  }
  if (!rsxDone)
'''

# large AlignedAllocator blocks -- the pixel store of
# RawImageData::createData(), common/RawImage.cpp:68-100 -- from a pool of page-locked memory
HUNK_ALLOC = r'''
    // rsx drop-in (optional, off by default): page-locked blocks for large buffers
    if (void* rsxP = rsx_shim::pool_alloc(numBytes))
      return static_cast<T*>(rsxP);
'''

HUNK_DEALLOC = r'''
    if (rsx_shim::pool_free(p)) // rsx drop-in: one of the pool's blocks
      return;
'''

PATCHES = [
    ("adt/AlignedAllocator.h", [
        # (behind the fuzzing build's 2 GB bail-out, in front of operator new)
        ("      ThrowRSE(\"FUZZ alloc bailout (%zu bytes)\", numBytes);\n#endif\n", HUNK_ALLOC),
        ("    invariant(isAligned(p, alignment));\n", HUNK_DEALLOC)], "rsx_pin.h"),
    ("decompressors/AbstractDngDecompressor.cpp", [
        ("void AbstractDngDecompressor::decompress() const {", HUNK_DNG)]),
    ("decompressors/UncompressedDecompressor.cpp", [
        ("void UncompressedDecompressor::readUncompressedRaw() {", HUNK_UNPACK),
        ("void UncompressedDecompressor::decode8BitRaw() {", HUNK_8BIT),
        ("void UncompressedDecompressor::decode12BitRawWithControl() {", HUNK_CONTROL),
        ("void UncompressedDecompressor::decode12BitRawUnpackedLeftAligned() {", HUNK_LEFT),
    ]),
    ("decompressors/NikonDecompressor.cpp", [
        ("  RawImageCurveGuard curveHandler(&mRaw, curve, uncorrectedRawValues);\n", HUNK_NIKON)]),
    ("decompressors/PentaxDecompressor.cpp", [
        ("void PentaxDecompressor::decompress(ByteStream data) const {", HUNK_PENTAX)]),
    ("decompressors/SamsungV1Decompressor.cpp", [
        ("void SamsungV1Decompressor::decompress() const {", HUNK_SAMSUNG_V1)]),
    ("decompressors/SamsungV2Decompressor.cpp", [
        ("void SamsungV2Decompressor::decompress() {", HUNK_SAMSUNG_V2)]),
    ("interpolators/Cr2sRawInterpolator.cpp", [
        ("void Cr2sRawInterpolator::interpolate(int version) {", HUNK_SRAW)]),
    ("decompressors/HasselbladDecompressor.cpp", [
        ("ByteStream::size_type HasselbladDecompressor::decompress() {", HUNK_HASSELBLAD)]),
    ("decompressors/SonyArw1Decompressor.cpp", [
        ("void SonyArw1Decompressor::decompress(ByteStream input) const {", HUNK_SONY_ARW1)]),
    ("decompressors/LJpegDecompressor.cpp", [
        ("ByteStream::size_type LJpegDecompressor::decode() const {", HUNK_LJPEG)]),
    ("decompressors/Cr2DecompressorImpl.h", [
        ("ByteStream::size_type Cr2Decompressor<PrefixCodeDecoder>::decompress() const {",
         HUNK_CR2)]),
]


def main():
    for entry in PATCHES:
        rel, hunks = entry[0], entry[1]
        include = entry[2] if len(entry) > 2 else "rsx_rawspeed_shim.h"
        orig = open(os.path.join(S, rel)).read()
        # Insertions, applied back to front so that the offsets stay those of the original.
        # Every insertion ends with a #line directive that restores the original line
        # numbering: the reference's exception texts embed __LINE__
        # (common/RawspeedException.h:83-86), and with the numbering intact a failure that
        # falls through to the original body reads the same as in the unmodified build.
        edits = []
        ns = orig.index("namespace rawspeed {")
        edits.append((ns, '#include "%s" // rsx drop-in\n#line %d\n'
                      % (include, orig.count("\n", 0, ns) + 1)))
        for anchor, hunk in hunks:
            if orig.count(anchor) != 1:
                raise SystemExit("anchor %r not found exactly once in %s" % (anchor, rel))
            end = orig.index(anchor) + len(anchor)
            if anchor.endswith("\n"):  # the anchor is whole lines: the hunk follows them
                text = hunk.lstrip("\n") + "#line %d\n" % (orig.count("\n", 0, end) + 1)
            else:                      # the anchor ends a line ("... {")
                text = hunk + "#line %d" % (orig.count("\n", 0, end) + 2)
            edits.append((end, text))
        src = orig
        for pos, text in sorted(edits, reverse=True):
            src = src[:pos] + text + src[pos:]
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(src)
        print("patched", rel)


if __name__ == "__main__":
    main()
