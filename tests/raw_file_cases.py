"""TEST INFRASTRUCTURE: the synthetic raw files of the decoder-level tests.
CASES[name]() -> (file bytes, expected uncropped image as (rows, width*cpp) uint16)."""
import numpy as np

from rawspeed_amd import abi, synth

import cases as C
import rawfiles


def _tiles(src, tw, th, pad=1000, slots=(0, 0), tables=(C.NIKON,), **kw):
    H, W = src.shape
    blobs = []
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            tile = np.full((th, tw), pad, np.uint16)
            part = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile[:part.shape[0], :part.shape[1]] = part
            blob, _, _, _ = synth.ljpeg_container(tile, 2, 14, list(slots), list(tables), **kw)
            blobs.append(blob)
    return blobs


def dng_ljpeg_tiles():
    """DngDecoder -> AbstractDngDecompressor::decompress<7>: 1021x700, 2x3 LJPEG tiles
    with overhang on the right and at the bottom."""
    rng = np.random.default_rng(501)
    W, H, tw, th = 1021, 700, 512, 256
    src = C.smooth_image(rng, H, W)
    return rawfiles.dng_file(W, H, tw, th, _tiles(src, tw, th)), src


def dng_ljpeg_tiles_two_tables():
    """What DNG writers emit: a Huffman table of its own per component (DHT slots 0 and 1)."""
    rng = np.random.default_rng(511)
    W, H, tw, th = 1021, 700, 512, 256
    src = C.smooth_image(rng, H, W)
    return rawfiles.dng_file(W, H, tw, th, _tiles(src, tw, th, slots=(0, 1),
                                                  tables=(C.NIKON, C.ALT))), src


def dng_ljpeg_tiles_dri():
    """The same with restart intervals inside every tile."""
    rng = np.random.default_rng(502)
    W, H, tw, th = 1024, 512, 512, 256
    src = C.smooth_image(rng, H, W)
    return rawfiles.dng_file(W, H, tw, th, _tiles(src, tw, th, rows_per_ri=32)), src


def dng_ljpeg_strips():
    """LJPEG in strips (one tile column: DngDecoder::getTilingDescription, strips branch)."""
    rng = np.random.default_rng(503)
    W, H, th = 768, 600, 200
    src = C.smooth_image(rng, H, W)
    return rawfiles.dng_file(W, H, W, th, _tiles(src, W, th), strips=True), src


def dng_uncompressed_12bit_strips():
    """DngDecoder -> decompress<1> -> UncompressedDecompressor (12-bit MSB packed)."""
    rng = np.random.default_rng(504)
    W, H, th, bps = 1024, 300, 128, 12
    src = rng.integers(0, 1 << bps, size=(H, W), dtype=np.uint16)
    blobs = [synth.pack_rows(src[y:y + th], bps, abi.ORDER_MSB) for y in range(0, H, th)]
    return rawfiles.dng_file(W, H, W, th, blobs, compression=1, bps=bps, strips=True), src


def dng_uncompressed_16bit_tiles():
    """16-bit little-endian tiles in several tile columns (the copyPixels path)."""
    rng = np.random.default_rng(505)
    W, H, tw, th = 1000, 300, 256, 128
    src = rng.integers(0, 1 << 16, size=(H, W), dtype=np.uint16)
    blobs = []
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            tile = np.zeros((th, tw), np.uint16)
            part = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile[:part.shape[0], :part.shape[1]] = part
            blobs.append(tile.view(np.uint8).reshape(-1))
    return rawfiles.dng_file(W, H, tw, th, blobs, compression=1, bps=16), src


def arw_ljpeg_tiles():
    """ArwDecoder::DecodeLJpeg (ArwDecoder.cpp:296-411): one LJpegDecoder per tile in an
    OpenMP loop, each decoding straight into the image."""
    rng = np.random.default_rng(506)
    W, H, tw, th = 1024, 768, 512, 256
    src = C.smooth_image(rng, H, W)
    return rawfiles.arw_file(W, H, tw, th, _tiles(src, tw, th)), src


def arw_uncompressed():
    """ArwDecoder::DecodeUncompressed: 16-bit LSB containers."""
    rng = np.random.default_rng(507)
    W, H = 1024, 512
    src = rng.integers(0, 1 << 14, size=(H, W), dtype=np.uint16)
    return rawfiles.arw_uncompressed_file(W, H, src.view(np.uint8).reshape(-1)), src


def arw1_compressed():
    """ArwDecoder -> SonyArw1Decompressor (compression 32767, strip size != w*h*bpp/8)."""
    rng = np.random.default_rng(508)
    W, H = 640, 488
    x = np.arange(W)[None, :]
    y = np.arange(H)[:, None]
    src = np.clip(900 + 900.0 * x / W + 700.0 * y / H + rng.normal(0, 12, (H, W)),
                  0, 4095).astype(np.uint16)
    data, _ = synth.sony_arw1_encode(src)
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    return rawfiles.arw1_file(W, H, data), src


def cr2_three_slices():
    """Cr2Decoder::decodeNewFormat -> Cr2LJpegDecoder -> Cr2Decompressor<2,1,1>."""
    rng = np.random.default_rng(509)
    W, H = 2016, 1100
    src = C.smooth_image(rng, H, W)
    rows = C.cr2_stream_from_image(src, 2, W // 2, H, [672, 672, 672])
    blob, _, _, _ = synth.ljpeg_container(rows, 2, 14, [0, 0], [C.NIKON])
    return rawfiles.cr2_file(W, H, blob, (2, 672, 672)), src


def pef_compressed():
    """PefDecoder -> PentaxDecompressor with the makernote's Huffman table."""
    import nikon_cases as N
    rng = np.random.default_rng(510)
    W, H = 1200, 300
    tree = synth.PENTAX_TREE
    src = N.smooth15(rng, H, W, maxv=4095, sigma=6.0)
    data, _ = N.pentax_encode(src, tree)
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    return rawfiles.pef_file(W, H, data, N.pentax_metadata(tree)), src


def _nef(seed, unc):
    import golden_cases as G
    import nikon_cases as N
    rng = np.random.default_rng(seed)
    W, H, bits = 1200, 300, 14
    p_up = [int(x) for x in rng.integers(1500, 2500, size=4)]
    pts = [] if unc else list(G.nikon_curve_points(300, 4000))
    meta = N.metadata(70, 0, p_up, pts, 0, pad_to=3000)
    P = N.parse(meta, bits, H)
    src = N.smooth15(rng, H, W, maxv=(1 << bits) - 1)
    pu = P["p_up"]
    data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                 synth.NIKON_TREE[P["huff_select"]])
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    return rawfiles.nef_file(W, H, bits, data, meta), src


def nef_compressed_uncorrected():
    """NefDecoder -> NikonDecompressor, uncorrectedRawValues: the image is the source."""
    return _nef(511, True)


def nef_compressed_curve():
    """... with the linearisation curve and its dither: no closed-form expectation here,
    the two builds of the reference must agree (the class-level tests pin the values)."""
    blob, _ = _nef(512, False)
    return blob, None


def threefr_ljpeg():
    """ThreefrDecoder -> HasselbladLJpegDecoder -> HasselbladDecompressor."""
    rng = np.random.default_rng(513)
    W, H = 1024, 300
    src = C.smooth_image(rng, H, W, 14)
    scan, _ = synth.hasselblad_encode(src, 0x2000, C.FULL17)
    hdr = synth.ljpeg_header(14, W, H, 1, [0], [C.FULL17])
    blob = np.concatenate([hdr, scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    return rawfiles.threefr_file(W, H, blob), src


def srw_samsung_v1():
    """SrwDecoder -> SamsungV1Decompressor (compression 32772)."""
    import nikon_cases as N
    rng = np.random.default_rng(514)
    W, H = 1024, 300
    src = N.smooth15(rng, H, W, maxv=4095, sigma=6.0)
    data, _ = synth.prefix_encode(src, [0, 0, 0, 0], synth.SAMSUNG_V1_TAB)
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    return rawfiles.srw_v1_file(W, H, data), src


def _cr2_sraw(seed, ysf):
    rng = np.random.default_rng(seed)
    groups = (3, 216, 160)
    gs, dim_x, dim_y = 2 + 2 * ysf, 2 * 216 + 160, 300
    # luma mid-range, chroma near its zero (16384): the RGB values stay inside 16 bits
    img = C.smooth_image(rng, dim_y, dim_x * gs, 13).astype(np.int64) + 2000
    chroma = np.zeros(dim_x * gs, bool)
    chroma[gs - 2::gs] = chroma[gs - 1::gs] = True
    img[:, chroma] = 16384 + (img[:, chroma] - 6000) // 16
    d, data, src, _, rows = C.make_cr2_sraw_case(rng, ysf, groups, dim_y, with_rows=True,
                                                 img=img.astype(np.uint16))
    blob, _, _, _ = synth.ljpeg_container(
        rows, 3, 14, [0, 0, 0], [C.NIKON], frame_wh=(d.frame_w, d.frame_h),
        samp=[(2, ysf), (1, 1), (1, 1)], pattern=synth.SRAW_PATTERN[2 + 2 * ysf])
    unit = 4 if ysf == 1 else 6
    h, w = src.shape                      # samples per row = groups * (2 + 2 ysf)
    sensor_w = w // (2 + 2 * ysf) * 2     # dim.x = sensor_w / 2 * (2 + 2 ysf)
    return rawfiles.cr2_sraw_file(sensor_w, h * ysf, blob,
                                  (2, groups[1] * unit, groups[2] * unit), ysf), None


def cr2_sraw_2x1():
    """Cr2Decoder sRaw 4:2:2: Cr2Decompressor<3,2,1>, then Cr2sRawInterpolator -> RGB."""
    return _cr2_sraw(515, 1)


def cr2_sraw_2x2():
    """... 4:2:0: Cr2Decompressor<3,2,2> and the two-row interpolation."""
    return _cr2_sraw(516, 2)


def srw_samsung_v2():
    """SrwDecoder -> SamsungV2Decompressor (compression 32773): NOT forwarded to the GPU --
    the patched build must decode it with the reference's own code, untouched."""
    import samsung_v2_cases as V2
    rng = np.random.default_rng(517)
    W, H, bits = 256, 60, 12
    x = np.arange(W)[None, :]
    y = np.arange(H)[:, None]
    target = np.clip(1200 + 900.0 * x / W + 600.0 * y / H + rng.normal(0, 12, (H, W)),
                     0, 4095).astype(np.int64)
    data, want = V2.encode(rng, target, bits, optflags=0)
    return rawfiles.srw_v1_file(W, H, data, bits=bits, compression=32773), want


# decodeRaw() options of a case (default: corrected values)
UNCORRECTED = {"nef_compressed_uncorrected"}

CASES = {f.__name__: f for f in (
    dng_ljpeg_tiles, dng_ljpeg_tiles_two_tables, dng_ljpeg_tiles_dri, dng_ljpeg_strips, dng_uncompressed_12bit_strips,
    dng_uncompressed_16bit_tiles, arw_ljpeg_tiles, arw_uncompressed, arw1_compressed,
    cr2_three_slices, pef_compressed, nef_compressed_uncorrected, nef_compressed_curve,
    threefr_ljpeg, srw_samsung_v1, cr2_sraw_2x1, cr2_sraw_2x2, srw_samsung_v2)}
