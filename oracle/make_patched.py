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

HUNK_UNPACK = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1) ----
  if (mRaw->getDataType() == RawImageType::F32) {
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
    if (int st = rsx_unpack_f32(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    return;
  }
  if (mRaw->getDataType() == RawImageType::UINT16) {
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
    if (int st = rsx_unpack_u16(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    return;
  }
'''

HUNK_VARIANT = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1b) ----
  {
    rsx_unpack_variant_desc d{};
    d.variant = %(variant)s;
    d.big_endian = %(big)s;
    d.w = size.x;
    d.h = size.y;
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    if (int st = rsx_unpack_variant_u16(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    input.skipBytes(input.getRemainSize());
    return;
  }
'''

HUNK_8BIT = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 1b) ----
  {
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
    if (int st = rsx_unpack_variant_u16(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    input.skipBytes(input.getRemainSize());
    return;
  }
'''
HUNK_CONTROL = HUNK_VARIANT % dict(variant="RSX_UNPACK_12BIT_WITH_CONTROL",
                                   big="e == Endianness::big")
HUNK_LEFT = HUNK_VARIANT % dict(variant="RSX_UNPACK_12BIT_UNPACKED_LEFT_ALIGNED",
                                big="e == Endianness::big")

HUNK_LJPEG = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 2) ----
  {
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
    const rsx_image img = rsx_shim::view(mRaw);
    uint32_t consumed = 0;
    if (int st = rsx_ljpeg_decode(rsx_shim::context(), &d, input.begin(),
                                  implicit_cast<size_t>(input.size()), &img, &consumed))
      rsx_shim::raise(st);
    return consumed;
  }
'''

HUNK_CR2 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3) ----
  {
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
    if (int st = rsx_cr2_decode(rsx_shim::context(), &d, input.begin(),
                                implicit_cast<size_t>(input.size()), &img, &consumed))
      rsx_shim::raise(st);
    return consumed;
  }
'''

HUNK_NIKON = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3b) ----
  {
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
    if (int st = rsx_nikon_decompress(rsx_shim::context(), &d, input.begin(),
                                      implicit_cast<size_t>(input.size()), &img))
      rsx_shim::raise(st);
    // what ~RawImageCurveGuard leaves behind (common/RawImage.h:376-382)
    if (uncorrectedRawValues)
      mRaw->setTable(curve, false);
    else
      mRaw->setTable(nullptr);
    return;
  }
'''

HUNK_PENTAX = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3c) ----
  {
    rsx_pentax_desc d{};
    d.table = rsx_shim::table(ht);
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = data.peekRemainingBuffer();
    if (int st = rsx_pentax_decompress(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    return;
  }
'''

HUNK_SAMSUNG_V1 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3d) ----
  {
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
    if (int st = rsx_samsung_v1_decompress(rsx_shim::context(), &d, in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    return;
  }
'''

HUNK_SRAW = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3a) ----
  {
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
    if (int st = rsx_sraw_interpolate(rsx_shim::context(), &d, &in, &out))
      rsx_shim::raise(st);
    return;
  }
'''

HUNK_SONY_ARW1 = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3f) ----
  {
    const rsx_image img = rsx_shim::view(mRaw);
    const Buffer in = input.peekRemainingBuffer();
    if (int st = rsx_sony_arw1_decompress(rsx_shim::context(), in.begin(), in.getSize(), &img))
      rsx_shim::raise(st);
    return;
  }
'''

HUNK_HASSELBLAD = r'''
  // ---- rsx: forward to the MI355X core (INTEGRATION.md 3e) ----
  {
    rec.ht.verifyCodeValuesAsDiffLengths();
    rsx_hasselblad_desc d{};
    d.table = rsx_shim::table(rec.ht);
    d.init_pred = rec.initPred;
    const rsx_image img = rsx_shim::view(mRaw);
    uint32_t consumed = 0;
    if (int st = rsx_hasselblad_decompress(rsx_shim::context(), &d, input.begin(),
                                           implicit_cast<size_t>(input.size()), &img, &consumed))
      rsx_shim::raise(st);
    return consumed;
  }
'''

PATCHES = [
    ("decompressors/UncompressedDecompressor.cpp", [
        ("void UncompressedDecompressor::readUncompressedRaw() {", HUNK_UNPACK),
        ("void UncompressedDecompressor::decode8BitRaw() {", HUNK_8BIT),
        ("void UncompressedDecompressor::decode12BitRawWithControl() {", HUNK_CONTROL),
        ("void UncompressedDecompressor::decode12BitRawUnpackedLeftAligned() {", HUNK_LEFT),
    ]),
    ("decompressors/NikonDecompressor.cpp", [
        ("void NikonDecompressor::decompress(Array1DRef<const uint8_t> input,\n"
         "                                   bool uncorrectedRawValues) {", HUNK_NIKON)]),
    ("decompressors/PentaxDecompressor.cpp", [
        ("void PentaxDecompressor::decompress(ByteStream data) const {", HUNK_PENTAX)]),
    ("decompressors/SamsungV1Decompressor.cpp", [
        ("void SamsungV1Decompressor::decompress() const {", HUNK_SAMSUNG_V1)]),
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
    for rel, hunks in PATCHES:
        src = open(os.path.join(S, rel)).read()
        # the shim include goes after the file's last #include
        last_inc = src.rfind("#include ")
        eol = src.index("\n", last_inc) + 1
        src = src[:eol] + '#include "rsx_rawspeed_shim.h" // rsx drop-in\n' + src[eol:]
        for anchor, hunk in hunks:
            if src.count(anchor) != 1:
                raise SystemExit("anchor %r not found exactly once in %s" % (anchor, rel))
            src = src.replace(anchor, anchor + hunk, 1)
        dst = os.path.join(OUT, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(dst, "w") as f:
            f.write(src)
        print("patched", rel)


if __name__ == "__main__":
    main()
