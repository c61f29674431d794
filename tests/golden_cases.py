"""The seeded case list behind tests/golden/*.json.

Each case is rebuilt deterministically from its parameters (numpy Generator
PCG64 + librsx_synth), decoded, and hashed rstest-style: md5 of the
concatenated per-line md5 hex digests of the uncropped image, padding excluded
(src/utilities/rstest/rstest.cpp:131-146).  tests/golden/make_golden.py stores
the hashes produced by the *reference build* (oracle/_ref); the tests then hold
the oracle -- and, on the GPU, the HIP path -- to those hashes.
"""
import hashlib

import numpy as np

from rawspeed_amd import abi, synth

import cases as C


def image_hash(pixels):
    """pixels: (rows, dim_x*cpp) uint16 view without row padding."""
    lines = "".join(hashlib.md5(np.ascontiguousarray(r).tobytes()).hexdigest()
                    for r in pixels)
    return hashlib.md5(lines.encode()).hexdigest()


UNPACK_CASES = []
for order in range(4):
    for bps in (1, 4, 7, 8, 10, 12, 13, 14, 15, 16):
        for (w, h, pad, oy) in ((48, 9, 0, 0), (80, 5, 3, 2), (2056, 3, 1, 0)):
            if (w * bps) % 8:
                continue
            UNPACK_CASES.append(dict(order=order, bps=bps, w=w, h=h, pad=pad, oy=oy))


def build_unpack(c, seed=1234):
    rng = np.random.default_rng([seed, c["order"], c["bps"], c["w"], c["pad"]])
    pitch = c["w"] * c["bps"] // 8 + c["pad"]
    data = rng.integers(0, 256, size=c["h"] * pitch, dtype=np.uint8)
    d = abi.UnpackDesc(0, c["oy"], c["w"], c["h"], pitch, c["bps"], c["order"])
    return d, data, (c["w"], c["h"] + c["oy"], 1)


# decode8BitRaw<true> / decode12BitRawWithControl<e> /
# decode12BitRawUnpackedLeftAligned<e>: (variant, big_endian, w, h, extra input bytes)
VARIANT_CASES = []
for variant in range(3):
    for big in ((0,) if variant == 0 else (0, 1)):
        for (w, h, extra) in ((10, 3, 0), (48, 5, 0), (62, 4, 7), (2568, 3, 0),
                              (5126, 2, 1)):
            VARIANT_CASES.append(dict(variant=variant, big=big, w=w, h=h, extra=extra))


def variant_bpl(variant, w):
    return (w, 12 * w // 8 + (w + 2) // 10, 2 * w)[variant]


def build_variant(c, seed=4321):
    rng = np.random.default_rng([seed, c["variant"], c["big"], c["w"], c["h"]])
    n = variant_bpl(c["variant"], c["w"]) * c["h"] + c["extra"]
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    d = abi.UnpackVariantDesc(c["variant"], c["big"], c["w"], c["h"])
    return d, data, (c["w"], c["h"], 1)


# F32 images: (order, bps, w, h, cpp, pad, ox, oy); special values (zero, subnormals,
# inf, NaN payloads) are forced into the first samples
F32_CASES = []
for order, bps in ((0, 16), (1, 16), (0, 24), (1, 24), (0, 32), (1, 32), (2, 32)):
    for (w, h, cpp, pad, ox, oy) in ((40, 6, 1, 0, 0, 0), (30, 5, 2, 3, 2, 1), (1028, 3, 3, 1, 0, 0)):
        F32_CASES.append(dict(order=order, bps=bps, w=w, h=h, cpp=cpp, pad=pad, ox=ox, oy=oy))


def build_f32(c, seed=99):
    rng = np.random.default_rng([seed, c["order"], c["bps"], c["w"], c["cpp"]])
    nbytes = c["bps"] // 8
    row = c["w"] * c["cpp"] * nbytes
    pitch = row + c["pad"]
    data = rng.integers(0, 256, size=c["h"] * pitch, dtype=np.uint8)
    # specials: exponent all-zero / all-one with zero and non-zero fractions
    specials = {16: [0x0000, 0x8000, 0x0001, 0x83FF, 0x7C00, 0xFC00, 0x7C01, 0xFFFF, 0x0400],
                24: [0x000000, 0x800000, 0x000001, 0x80FFFF, 0x7F0000, 0xFF0000, 0x7F0001,
                     0xFFFFFF, 0x010000],
                32: [0, 0x80000000, 1, 0x7F800000, 0x7FC00001]}[c["bps"]]
    for i, v in enumerate(specials):
        b = int(v).to_bytes(nbytes, "big" if c["order"] == 1 and c["bps"] != 32 else "little")
        data[i * nbytes:(i + 1) * nbytes] = np.frombuffer(b, np.uint8)
    d = abi.UnpackDesc(c["ox"], c["oy"], c["w"], c["h"], pitch, c["bps"], c["order"])
    return d, data, (c["w"] + c["ox"] + 1, c["h"] + c["oy"], c["cpp"])


LJPEG_CASES = [
    dict(name="mcu2x1_full", img=(64, 8, 1), tile=(0, 0, 64, 8), mcu=(2, 1)),
    dict(name="mcu1x1", img=(64, 8, 1), tile=(0, 0, 64, 8), mcu=(1, 1)),
    dict(name="mcu3x1", img=(63, 8, 1), tile=(0, 0, 63, 8), mcu=(3, 1)),
    dict(name="mcu4x1", img=(64, 8, 1), tile=(0, 0, 64, 8), mcu=(4, 1)),
    dict(name="mcu2x2", img=(64, 8, 1), tile=(0, 0, 64, 8), mcu=(2, 2)),
    dict(name="overhang_br", img=(29, 13, 1), tile=(16, 8, 13, 5), mcu=(2, 1), frame=(8, 8)),
    dict(name="overhang_tl", img=(29, 13, 1), tile=(0, 0, 16, 8), mcu=(2, 1), frame=(8, 8)),
    dict(name="cpp3", img=(20, 12, 3), tile=(2, 3, 10, 6), mcu=(3, 1)),
    dict(name="cpp2_wideframe", img=(20, 12, 2), tile=(2, 2, 9, 6), mcu=(4, 1), frame=(6, 7)),
    dict(name="dri3", img=(64, 12, 1), tile=(0, 0, 64, 12), mcu=(2, 1), rows_per_ri=3),
    dict(name="dri5", img=(64, 12, 1), tile=(0, 0, 64, 12), mcu=(2, 1), rows_per_ri=5),
    dict(name="two_tables", img=(64, 16, 1), tile=(0, 0, 64, 16), mcu=(2, 1),
         tables=("NIKON", "ALT"), table_index=[0, 1]),
    dict(name="full17_16bit", img=(64, 16, 1), tile=(0, 0, 64, 16), mcu=(2, 1),
         tables=("FULL17",), full_range=True, prec=16),
    dict(name="full17_16bit_fix16", img=(64, 16, 1), tile=(0, 0, 64, 16), mcu=(2, 1),
         tables=("FULL17",), full_range=True, prec=16, fix16=True),
    dict(name="medium", img=(640, 160, 1), tile=(0, 0, 640, 160), mcu=(2, 1)),
    dict(name="medium_2x2_tables", img=(320, 128, 1), tile=(0, 0, 320, 128), mcu=(2, 2),
         tables=("NIKON", "ALT", "FULL17"), table_index=[0, 1, 2, 1]),
    dict(name="large", img=(2048, 512, 1), tile=(0, 0, 2048, 512), mcu=(2, 1)),
]

TABLES = {"NIKON": C.NIKON, "FULL17": C.FULL17, "ALT": C.ALT}


def build_ljpeg(c, seed=4321):
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    kw = {k: v for k, v in c.items() if k not in ("name", "img", "tables")}
    kw["tables"] = tuple(TABLES[t] for t in c.get("tables", ("NIKON",)))
    w, h, cpp = c["img"]
    d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=cpp, **kw)
    return d, data, (w, h, cpp), tile_px


CR2_CASES = [
    dict(name="one_slice", w=64, h=8, n=2, slices=(1, 0, 64)),
    dict(name="three_slices", w=40, h=24, n=2, slices=(3, 16, 8)),
    dict(name="four_comp", w=48, h=24, n=4, slices=(3, 16, 16)),
    dict(name="four_comp_tables", w=48, h=24, n=4, slices=(2, 32, 16),
         tables=("NIKON", "ALT"), table_index=[0, 1, 1, 0]),
    dict(name="medium", w=672, h=448, n=2, slices=(3, 224, 224)),
    dict(name="large", w=2016, h=640, n=2, slices=(3, 672, 672)),
    # Canon sRaw: slices in groups, image = sum(widths) * (2 + 2*ysf) samples wide
    dict(name="sraw_321_one_slice", sraw=1, slices=(1, 0, 24), h=10),
    dict(name="sraw_321_slices", sraw=1, slices=(3, 40, 24), h=36),
    dict(name="sraw_322_slices", sraw=2, slices=(3, 32, 20), h=30),
    dict(name="sraw_322_tables", sraw=2, slices=(2, 48, 16), h=24,
         tables=("NIKON", "ALT"), table_index=[0, 1, 1]),
    dict(name="sraw_321_wrapped", sraw=1, slices=(4, 30, 30), h=20, dim_x=60, frame_y=10),
    dict(name="sraw_322_wrapped_partial", sraw=2, slices=(5, 16, 16), h=21, dim_x=32,
         frame_y=9),
    dict(name="sraw_322_medium", sraw=2, slices=(3, 160, 128), h=300),
]


def build_cr2(c, seed=777):
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    kw = {}
    if "tables" in c:
        kw["tables"] = tuple(TABLES[t] for t in c["tables"])
        kw["table_index"] = c["table_index"]
    if "sraw" in c:
        for k in ("dim_x", "frame_y"):
            if k in c:
                kw[k] = c[k]
        d, data, img, scan_len = C.make_cr2_sraw_case(rng, c["sraw"], c["slices"], c["h"], **kw)
        return d, data, (img.shape[1], c["h"], 1), img
    d, data, img, scan_len = C.make_cr2_case(rng, c["w"], c["h"], c["n"], c["slices"], **kw)
    return d, data, (c["w"], c["h"], 1), img


# ---- NikonDecompressor -------------------------------------------------------
# v0/v1 pick the metadata flavour (NikonDecompressor.cpp:493-513, createCurve):
# 70 = lossless (identity curve), 68/32|64 = lossy with an interpolated curve and
# a split row, anything else = a literal curve.  kind "image": a smooth 15-bit
# image encoded with tree huffSelect (round trip when uncorrected); "symbols":
# random symbols (the only way to exercise the "after split" trees).
NIKON_CASES = [
    dict(name="lossless14_image", v0=70, v1=0, bits=14, w=64, h=20, kind="image", unc=1),
    dict(name="lossless14_dither", v0=70, v1=0, bits=14, w=64, h=20, kind="image", unc=0),
    dict(name="lossless12_image", v0=70, v1=0, bits=12, w=48, h=18, kind="image", unc=1),
    dict(name="lossy12_curve", v0=68, v1=32, bits=12, w=40, h=16, kind="symbols", unc=0),
    dict(name="lossy12_split", v0=68, v1=32, bits=12, w=40, h=16, split=7, kind="symbols", unc=0),
    dict(name="lossy12_split_unc", v0=68, v1=32, bits=12, w=40, h=16, split=7, kind="symbols", unc=1),
    dict(name="lossy14_z7_split", v0=68, v1=64, bits=14, w=48, h=20, split=9, kind="symbols", unc=0),
    dict(name="lossy14_curve", v0=68, v1=32, bits=14, w=48, h=12, kind="symbols", unc=0),
    dict(name="literal_curve", v0=68, v1=16, bits=12, w=40, h=10, kind="symbols", unc=0),
    dict(name="skip2110", v0=73, v1=0, bits=12, w=32, h=8, kind="symbols", unc=0),
    dict(name="lossless_v1_88", v0=70, v1=88, bits=14, w=32, h=8, kind="symbols", unc=0),
    dict(name="split_outside", v0=68, v1=32, bits=12, w=40, h=16, split=16, kind="symbols", unc=0),
    dict(name="medium_image", v0=70, v1=0, bits=14, w=1200, h=300, kind="image", unc=0),
    dict(name="medium_split", v0=68, v1=32, bits=12, w=800, h=200, split=77, kind="symbols", unc=0),
]


def nikon_curve_points(n, maxv):
    x = np.linspace(0, 1, n)
    c = (maxv * x ** 0.8).astype(int)
    c[5:8] = c[5:8][::-1]  # a non-monotonic kink (TableLookUp.cpp:71-73)
    return c


def build_nikon(c, seed=2024):
    import nikon_cases as N
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    v0, v1, bits, w, h = c["v0"], c["v1"], c["bits"], c["w"], c["h"]
    if v0 == 68 and v1 in (32, 64):
        pts = nikon_curve_points(257, (1 << (bits - 2 if v1 == 64 else bits)) - 1)
    elif v0 != 70:
        pts = nikon_curve_points(300, 4000)
    else:
        pts = []
    p_up = [int(x) for x in rng.integers(1500, 2500, size=4)]
    meta = N.metadata(v0, v1, p_up, pts, c.get("split", 0), pad_to=3000)
    P = N.parse(meta, bits, h)
    hs = P["huff_select"]
    src = None
    if c["kind"] == "image":
        src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1)
        pu = P["p_up"]
        data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                     synth.NIKON_TREE[hs])
        data = np.concatenate([data, np.zeros(8, np.uint8)])
    else:
        sp = P["split"]
        n0 = (sp or h) * w
        n1 = (h - sp) * w if sp else 0
        data = N.symbol_stream(rng, n0, synth.NIKON_TREE[hs], n1,
                               synth.NIKON_TREE[hs + 1] if n1 else None)
    d = N.desc(P, bits, bool(c["unc"]))
    return meta, d, data, (w, h, 1), src


# ---- PentaxDecompressor ----------------------------------------------------------
# tree: "legacy" = pentax_tree (always handed over as a makernote blob: the
# reference's legacy branch is not reachable in our -O3 build of it, its
# Optional::has_value() is declared readnone), "modern" = a 15-symbol tree.
PENTAX_CASES = [
    dict(name="legacy_small", tree="legacy", w=64, h=20, maxv=4095),
    dict(name="modern_small", tree="modern", w=48, h=17, maxv=16383),
    dict(name="legacy_medium", tree="legacy", w=1200, h=300, maxv=4095),
    dict(name="modern_wide", tree="modern", w=8384, h=12, maxv=16383),
    dict(name="range_error", tree="legacy", w=64, h=20, maxv=4095, symbols=True),
]


def build_pentax(c, seed=515):
    import nikon_cases as N
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    tree = synth.PENTAX_TREE if c["tree"] == "legacy" else N.PENTAX_MODERN
    meta = N.pentax_metadata(tree)
    w, h = c["w"], c["h"]
    src = None
    if c.get("symbols"):
        data = N.symbol_stream(rng, w * h, tree)  # a random walk: leaves [0, 65535] quickly
    else:
        src = N.smooth15(rng, h, w, maxv=c["maxv"], sigma=6.0)
        data, _ = N.pentax_encode(src, tree)
        data = np.concatenate([data, np.zeros(8, np.uint8)])
    return meta, N.pentax_desc(tree), data, (w, h, 1), src


# ---- SamsungV1Decompressor -------------------------------------------------------
SAMSUNG_V1_CASES = [
    dict(name="small", w=64, h=20),
    dict(name="medium", w=1024, h=300),
    dict(name="max_width", w=5664, h=12),
    dict(name="range_error", w=64, h=20, symbols=True),
]


def build_samsung_v1(c, seed=616):
    import nikon_cases as N
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    w, h = c["w"], c["h"]
    d = abi.SamsungV1Desc.make(synth.SAMSUNG_V1_TAB)
    src = None
    if c.get("symbols"):
        data = rng.integers(0, 256, size=w * h * 2, dtype=np.uint8)  # any bits parse
    else:
        src = N.smooth15(rng, h, w, maxv=4095, sigma=6.0)
        data, _ = synth.prefix_encode(src, [0, 0, 0, 0], synth.SAMSUNG_V1_TAB)
        data = np.concatenate([data, np.zeros(8, np.uint8)])
    return d, data, (w, h, 1), src


# ---- SonyArw1Decompressor ------------------------------------------------------------
# (the decoder of the A100: 3881 x 2608, ArwDecoder.cpp:128-129; width may be odd)
SONY_ARW1_CASES = [
    dict(name="small", w=37, h=20, sigma=5.0),
    dict(name="one_column", w=1, h=2, sigma=1.0),
    dict(name="medium_odd_width", w=485, h=326, sigma=12.0),
    dict(name="noisy_long_codes", w=200, h=100, sigma=700.0),
    dict(name="tall", w=40, h=3072, sigma=8.0),
    dict(name="wide", w=4600, h=6, sigma=8.0),
    dict(name="range_error", w=64, h=20, sigma=5.0, poison=(7, 33, 5000)),
    dict(name="negative_value", w=64, h=20, sigma=5.0, poison=(12, 3, -3)),
    dict(name="random_bits", w=64, h=20, symbols=True),
]


def build_sony_arw1(c, seed=919):
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    w, h = c["w"], c["h"]
    if c.get("symbols"):
        return rng.integers(0, 256, size=w * h * 2, dtype=np.uint8), (w, h, 1), None
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    src = 900 + 900.0 * x / max(w, 1) + 700.0 * y / max(h, 1) + rng.normal(0, c["sigma"], (h, w))
    src = np.clip(src, 0, 4095).astype(np.int16)
    if c.get("poison"):                     # a value the decoder must reject
        r, col, v = c["poison"]
        src[r, col] = v
    data, _ = synth.sony_arw1_encode(src.view(np.uint16))
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    return data, (w, h, 1), src.view(np.uint16)


# ---- Cr2sRawInterpolator -----------------------------------------------------------
# (version, subsampling_y, groups per row, input rows); hue and white-balance
# coefficients in the range Cr2Decoder derives them (Cr2Decoder.cpp:560-625)
SRAW_CASES = [
    dict(name="422_v0", version=0, ysf=1, groups=24, rows=9),
    dict(name="422_v1", version=1, ysf=1, groups=40, rows=6),
    dict(name="422_v2", version=2, ysf=1, groups=2, rows=3),
    dict(name="420_v1", version=1, ysf=2, groups=24, rows=9),
    dict(name="420_v2", version=2, ysf=2, groups=40, rows=1),
    dict(name="420_v2_two_groups", version=2, ysf=2, groups=2, rows=2),
    dict(name="422_v2_medium", version=2, ysf=1, groups=1296, rows=200),
    dict(name="420_v1_medium", version=1, ysf=2, groups=990, rows=160),
    dict(name="420_v2_extreme", version=2, ysf=2, groups=64, rows=8, extreme=True),
]


def build_sraw(c, seed=717):
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    gs = 2 + 2 * c["ysf"]
    w = c["groups"] * gs
    if c.get("extreme"):   # any 16-bit input, huge coefficients: int wrap-around paths
        px = rng.integers(0, 65536, size=(c["rows"], w), dtype=np.uint16)
        coeffs = [65535, 40000, 65535]
        hue = 3000
    else:
        px = np.empty((c["rows"], w), np.uint16)
        g = px.reshape(c["rows"], c["groups"], gs)
        g[:, :, :gs - 2] = rng.integers(200, 15000, size=(c["rows"], c["groups"], gs - 2))
        g[:, :, gs - 2:] = rng.integers(16384 - 3000, 16384 + 3000, size=(c["rows"], c["groups"], 2))
        coeffs = [int(x) for x in rng.integers(800, 2600, size=3)]
        hue = int(rng.integers(-600, 600))
    d = abi.SrawDesc.make(c["version"], c["ysf"], coeffs, hue)
    return d, px, (w, c["rows"]), (2 * c["groups"], c["ysf"] * c["rows"])


# ---- HasselbladDecompressor -------------------------------------------------------
HASSELBLAD_CASES = [
    dict(name="small", w=64, h=20),
    dict(name="full_range", w=66, h=7, full=True),
    dict(name="medium", w=1200, h=300),
    dict(name="max_width", w=12000, h=6),
    dict(name="odd_tail", w=130, h=9),
]


def build_hasselblad(c, seed=818):
    rng = np.random.default_rng([seed, sum(map(ord, c["name"]))])
    w, h = c["w"], c["h"]
    if c.get("full"):
        src = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
        src[2, 4:8] = [0, 32768, 32768, 0]   # differences of exactly -32768
    else:
        src = C.smooth_image(rng, h, w, 14)
    init = 0x8000 if c.get("full") else 0x2000
    data, _ = synth.hasselblad_encode(src, init, C.FULL17)
    return abi.HasselbladDesc.make(C.FULL17, init), data, (w, h, 1), src
