"""Oracle (oracle/rsx_oracle.c) vs the golden hashes the reference produced
(tests/golden/golden_hashes.json, generator tests/golden/make_golden.py) and,
where oracle/_ref is present, vs the reference itself, full-buffer compare."""
import json
import os

import numpy as np
import pytest

import golden_cases as G
from oracle_lib import HostImage

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_hashes.json")) as f:
    GOLD = json.load(f)


@pytest.mark.parametrize("i", range(len(G.UNPACK_CASES)))
def test_unpack_golden(oracle, i):
    d, data, (w, h, cpp) = G.build_unpack(G.UNPACK_CASES[i])
    img = HostImage(w, h, cpp)
    st = oracle.unpack(d, data, img)
    g = GOLD["unpack"][str(i)]
    assert st == g["status"]
    assert G.image_hash(img.pixels()) == g["hash"]


@pytest.mark.parametrize("i", range(len(G.F32_CASES)))
def test_unpack_f32_golden(oracle, i):
    d, data, (w, h, cpp) = G.build_f32(G.F32_CASES[i])
    img = HostImage(w, h, cpp, bpc=4)
    st = oracle.unpack_f32(d, data, img)
    g = GOLD["f32"][str(i)]
    assert st == g["status"] == 0
    assert G.image_hash(img.u32()[:, :w * cpp]) == g["hash"]


@pytest.mark.parametrize("i", range(len(G.VARIANT_CASES)))
def test_unpack_variant_golden(oracle, i):
    d, data, (w, h, cpp) = G.build_variant(G.VARIANT_CASES[i])
    img = HostImage(w, h, cpp)
    st = oracle.unpack_variant(d, data, img)
    g = GOLD["variant"][str(i)]
    assert st == g["status"] == 0
    assert G.image_hash(img.pixels()) == g["hash"]


@pytest.mark.parametrize("c", G.LJPEG_CASES, ids=lambda c: c["name"])
def test_ljpeg_golden(oracle, c):
    d, data, (w, h, cpp), tile_px = G.build_ljpeg(c)
    img = HostImage(w, h, cpp)
    st, consumed = oracle.ljpeg(d, data, img)
    g = GOLD["ljpeg"][c["name"]]
    assert (st, consumed) == (g["status"], g["consumed"])
    assert G.image_hash(img.pixels()) == g["hash"]
    # round trip: the decoded tile is the image the stream was encoded from
    tx, ty, tw, th = c["tile"]
    assert np.array_equal(img.pixels()[ty:ty + th, cpp * tx:cpp * (tx + tw)], tile_px)


@pytest.mark.parametrize("c", G.CR2_CASES, ids=lambda c: c["name"])
def test_cr2_golden(oracle, c):
    d, data, (w, h, cpp), src = G.build_cr2(c)
    img = HostImage(w, h, cpp, is_cfa="sraw" not in c)
    st, consumed = oracle.cr2(d, data, img)
    g = GOLD["cr2"][c["name"]]
    assert (st, consumed) == (g["status"], g["consumed"])
    assert G.image_hash(img.pixels()) == g["hash"]
    assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("c", G.NIKON_CASES, ids=lambda c: c["name"])
def test_nikon_golden(oracle, c):
    meta, d, data, (w, h, cpp), src = G.build_nikon(c)
    img = HostImage(w, h, cpp)
    st = oracle.nikon(d, data, img)
    g = GOLD["nikon"][c["name"]]
    assert st == g["status"] == 0
    assert G.image_hash(img.pixels()) == g["hash"]
    if src is not None and c["unc"]:
        assert np.array_equal(img.pixels(), src)   # round trip


@pytest.mark.parametrize("c", G.PENTAX_CASES, ids=lambda c: c["name"])
def test_pentax_golden(oracle, c):
    meta, d, data, (w, h, cpp), src = G.build_pentax(c)
    img = HostImage(w, h, cpp)
    st = oracle.pentax(d, data, img)
    g = GOLD["pentax"][c["name"]]
    # the reference's "decoded value out of bounds" is a RawDecoderException
    # (status 1 in the shim); the C-ABI has a code of its own for it
    assert (st, g["status"]) in ((0, 0), (10, 1))
    if st == 0:
        assert G.image_hash(img.pixels()) == g["hash"]
        assert np.array_equal(img.pixels(), src)
    else:
        assert st == 10  # RSX_ERR_VALUE_RANGE <-> "decoded value out of bounds"


@pytest.mark.parametrize("c", G.SAMSUNG_V1_CASES, ids=lambda c: c["name"])
def test_samsung_v1_golden(oracle, c):
    d, data, (w, h, cpp), src = G.build_samsung_v1(c)
    img = HostImage(w, h, cpp)
    st = oracle.samsung_v1(d, data, img)
    g = GOLD["samsung_v1"][c["name"]]
    assert (st, g["status"]) in ((0, 0), (10, 1))   # range error = RawDecoderException
    if st == 0:
        assert G.image_hash(img.pixels()) == g["hash"]
        assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("c", G.SONY_ARW1_CASES, ids=lambda c: c["name"])
def test_sony_arw1_golden(oracle, c):
    data, (w, h, cpp), src = G.build_sony_arw1(c)
    img = HostImage(w, h, cpp)
    st = oracle.sony_arw1(data, img)
    g = GOLD["sony_arw1"][c["name"]]
    assert (st, g["status"]) in ((0, 0), (10, 1))   # range error = RawDecoderException
    if st == 0:
        assert G.image_hash(img.pixels()) == g["hash"]
        assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("c", G.SRAW_CASES, ids=lambda c: c["name"])
def test_sraw_golden(oracle, c):
    d, px, (iw, ih), (ow, oh) = G.build_sraw(c)
    src, dst = HostImage(iw, ih, 1, is_cfa=False), HostImage(ow, oh, 3, is_cfa=False)
    src.pixels()[:] = px
    assert oracle.sraw(d, src, dst) == GOLD["sraw"][c["name"]]["status"] == 0
    assert G.image_hash(dst.pixels()) == GOLD["sraw"][c["name"]]["hash"]


@pytest.mark.parametrize("c", G.HASSELBLAD_CASES, ids=lambda c: c["name"])
def test_hasselblad_golden(oracle, c):
    d, data, (w, h, cpp), src = G.build_hasselblad(c)
    img = HostImage(w, h, cpp)
    g = GOLD["hasselblad"][c["name"]]
    assert oracle.hasselblad(d, data, img) == (g["status"], g["consumed"]) == (0, g["consumed"])
    assert G.image_hash(img.pixels()) == g["hash"]
    assert np.array_equal(img.pixels(), src)


# ---- live cross-checks against the compiled reference ----------------------

def test_hasselblad_vs_ref(oracle, ref):
    """Full buffers, getStreamPosition(), and status parity at every cut (the
    MSB32 reader's partial last word and 8-byte position budget)."""
    for c in G.HASSELBLAD_CASES:
        d, data, (w, h, cpp), _ = G.build_hasselblad(c)
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        assert oracle.hasselblad(d, data, hi) == ref.hasselblad(d, data, ri), ref.last_error()
        assert np.array_equal(hi.u16(), ri.u16())
    d, data, (w, h, cpp), _ = G.build_hasselblad(G.HASSELBLAD_CASES[1])
    seen = set()
    for cut in list(range(1, 40)) + [60, 100, 300, len(data) // 2]:
        part = data[:len(data) - cut]
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.hasselblad(d, part, hi), ref.hasselblad(d, part, ri)
        assert so == sr, (cut, so, sr, ref.last_error())
        if so[0] == 0:
            assert np.array_equal(hi.u16(), ri.u16())
        seen.add(so[0])
    assert 0 in seen and len(seen) >= 2


@pytest.mark.parametrize("c", G.SRAW_CASES, ids=lambda c: c["name"])
def test_sraw_vs_ref(oracle, ref, c):
    d, px, (iw, ih), (ow, oh) = G.build_sraw(c)
    src, dst = HostImage(iw, ih, 1, is_cfa=False), HostImage(ow, oh, 3, is_cfa=False)
    src.pixels()[:] = px
    rsrc, rdst = ref.image(iw, ih, 1, False), ref.image(ow, oh, 3, False)
    rsrc.set_pixels(px)
    assert oracle.sraw(d, src, dst) == ref.sraw(d, rsrc, rdst) == 0, ref.last_error()
    assert np.array_equal(dst.u16(), rdst.u16())


def test_samsung_v1_vs_ref(oracle, ref):
    """Full buffers, and status parity at every cut of a truncated stream
    (samsungDiff refills with fill(23): a 9-bit later position budget)."""
    for c in G.SAMSUNG_V1_CASES:
        d, data, (w, h, cpp), _ = G.build_samsung_v1(c)
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.samsung_v1(d, data, hi), ref.samsung_v1(12, data, ri)
        assert (so, sr) in ((0, 0), (10, 1)), (c["name"], so, sr, ref.last_error())
        if so == 0:
            assert np.array_equal(hi.u16(), ri.u16())
    d, data, (w, h, cpp), _ = G.build_samsung_v1(G.SAMSUNG_V1_CASES[0])
    full = len(data) - 8
    seen = set()
    for cut in range(0, 40):
        part = data[:full - cut]
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.samsung_v1(d, part, hi), ref.samsung_v1(12, part, ri)
        assert so == sr or (so, sr) == (10, 1), (cut, so, sr, ref.last_error())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2


def test_sony_arw1_vs_ref(oracle, ref):
    """Every case against the reference, then status parity at every cut of a
    truncated stream (fill(32) per pixel: the plain 8-byte over-read budget)."""
    for c in G.SONY_ARW1_CASES:
        data, (w, h, cpp), src = G.build_sony_arw1(c)
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.sony_arw1(data, hi), ref.sony_arw1(data, ri)
        assert (so, sr) in ((0, 0), (10, 1)), (c["name"], so, sr, ref.last_error())
        if so == 0:
            assert np.array_equal(hi.u16(), ri.u16())
            assert np.array_equal(hi.pixels(), src)
        else:
            assert "Error decompressing" in ref.last_error()
        assert (so != 0) == bool(c.get("poison") or c.get("symbols")), c["name"]
    data, (w, h, cpp), _ = G.build_sony_arw1(G.SONY_ARW1_CASES[0])
    full = len(data) - 8
    seen = set()
    for cut in range(0, 40):
        part = data[:full - cut]
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.sony_arw1(part, hi), ref.sony_arw1(part, ri)
        assert so == sr or (so, sr) == (10, 1), (cut, so, sr, ref.last_error())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2
    # the constructor's checks (.cpp:39-51); an all-zero stream never decodes, so the
    # reference always throws -- in the constructor for shapes it rejects
    for w, h, cpp in [(4, 3, 1), (4601, 2, 1), (4, 3074, 1), (4, 2, 2), (4600, 3072, 1),
                      (1, 2, 1)]:
        ri = ref.image(w, h, cpp)
        assert ref.sony_arw1(np.zeros(64, np.uint8), ri) != 0
        rejected = "Unexpected" in ref.last_error()
        assert (oracle.sony_arw1_validate(HostImage(w, h, cpp)) != 0) == rejected, (w, h, cpp)


@pytest.mark.parametrize("c", G.PENTAX_CASES, ids=lambda c: c["name"])
def test_pentax_vs_ref(oracle, ref, c):
    """Also pins nikon_cases.pentax_metadata: the reference rebuilds the same
    Huffman table from the blob (SetupPrefixCodeDecoder_Modern)."""
    meta, d, data, (w, h, cpp), _ = G.build_pentax(c)
    hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
    so, sr = oracle.pentax(d, data, hi), ref.pentax(meta, data, ri)
    # the reference reports the range error as a RawDecoderException (status 1)
    assert (so, sr) in ((0, 0), (10, 1)), (so, sr, ref.last_error())
    if so == 0:
        assert np.array_equal(hi.u16(), ri.u16())
    else:
        assert "out of bounds" in ref.last_error()


@pytest.mark.parametrize("c", G.NIKON_CASES, ids=lambda c: c["name"])
def test_nikon_vs_ref(oracle, ref, c):
    """Full-buffer compare; also pins tests/nikon_cases.parse (the constructor
    restatement that fills rsx_nikon_desc) against the reference's own parse."""
    meta, d, data, (w, h, cpp), _ = G.build_nikon(c)
    for unc in (True, False):
        d.uncorrected_raw_values = 1 if unc else 0
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        assert oracle.nikon(d, data, hi) == ref.nikon(meta, c["bits"], data, ri, unc) == 0
        assert np.array_equal(hi.u16(), ri.u16())


def test_nikon_truncated_vs_ref(oracle, ref):
    """BitStreamerMSB's position budget: zeros are read for 8 bytes past the end,
    then IOException (BitStreamer.h:120-131) -- status parity at every cut."""
    import nikon_cases as N
    c = dict(name="trunc", v0=68, v1=32, bits=12, w=40, h=16, split=7, kind="symbols", unc=1)
    meta, d, data, (w, h, cpp), _ = G.build_nikon(c)
    full = int(np.flatnonzero(data)[-1]) + 1
    seen = set()
    for cut in list(range(0, 40)) + [60, 100, 200, full // 2]:
        n = full - cut
        if n < 1:
            continue
        part = data[:n]
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.nikon(d, part, hi), ref.nikon(meta, 12, part, ri, True)
        assert so == sr, (cut, so, sr, ref.last_error())
        if so == 0:
            assert np.array_equal(hi.u16(), ri.u16())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2


def test_decode8bit_lookup_vs_ref(oracle, ref):
    """decode8BitRaw<false>: the curve/dither flavour is a pure table lookup
    because its random state starts at 0 and 15700 * 0 + 0 == 0."""
    from rawspeed_amd import abi
    from oracle_lib import dither_lut8
    rng = np.random.default_rng(8)
    for n_curve in (256, 100, 1):
        curve = np.sort(rng.integers(0, 65536, size=n_curve)).astype(np.uint16)
        if n_curve > 10:
            curve[3:6] = curve[3:6][::-1]
        for (w, h) in ((16, 3), (250, 4)):
            data = rng.integers(0, 256, size=w * h, dtype=np.uint8)
            d = abi.UnpackVariantDesc(abi.UNPACK_8BIT_LOOKUP, 0, w, h).set_lut(dither_lut8(curve))
            a, b = HostImage(w, h, 1), ref.image(w, h, 1)
            assert oracle.unpack_variant(d, data, a) == 0
            assert ref.decode8bit_lookup(curve, w, h, data, b) == 0, ref.last_error()
            assert np.array_equal(a.u16(), b.u16())


def test_unpack_f32_vs_ref_sweep(oracle, ref):
    """F32 images: every (order, bps) the reference accepts or rejects, random bit
    patterns (all exponent classes), crops, paddings; compared as raw bits."""
    from rawspeed_amd import abi
    rng = np.random.default_rng(6)
    n_ok = n_err = 0
    for order in range(4):
        for bps in (8, 16, 24, 32):
            for (w, h, cpp, pad, ox, oy) in ((4, 1, 1, 0, 0, 0), (12, 3, 1, 0, 0, 0),
                                             (10, 4, 2, 5, 3, 1), (9, 2, 3, 1, 0, 2)):
                pitch = w * cpp * bps // 8 + pad
                for cut in (0, 1):
                    n = h * pitch - cut
                    data = rng.integers(0, 256, size=max(n, 1), dtype=np.uint8)[:n]
                    d = abi.UnpackDesc(ox, oy, w, h, pitch, bps, order)
                    dim_y = h + oy - int(rng.integers(0, 2))
                    a = HostImage(w + ox, max(dim_y, 1), cpp, bpc=4)
                    b = ref.image(w + ox, max(dim_y, 1), cpp, f32=True)
                    sa, sb = oracle.unpack_f32(d, data, a), ref.unpack(d, data, b)
                    assert sa == sb, (order, bps, w, h, cpp, pad, ox, oy, cut, sa, sb,
                                      ref.last_error())
                    assert np.array_equal(a.u32(), b.u32())
                    n_ok += sa == 0
                    n_err += sa != 0
    assert n_ok >= 20 and n_err >= 40


def test_unpack_variant_vs_ref_sweep(oracle, ref):
    """decode8BitRaw<true>, decode12BitRawWithControl<e>,
    decode12BitRawUnpackedLeftAligned<e> against the reference build: every
    width class mod 10, truncated inputs, full-buffer compare (padding too)."""
    from rawspeed_amd import abi
    rng = np.random.default_rng(5)
    n_ok = n_err = 0
    for variant in range(3):
        for big in (0, 1):
            for w in list(range(2, 44, 2)) + [250, 1002]:
                for h in (1, 3):
                    bpl = G.variant_bpl(variant, w)
                    for cut in (0, 0, 1, bpl):
                        n = bpl * h - cut
                        if n <= 0:
                            continue
                        data = rng.integers(0, 256, size=n, dtype=np.uint8)
                        d = abi.UnpackVariantDesc(variant, big, w, h)
                        a, b = HostImage(w, h, 1), ref.image(w, h, 1)
                        sa = oracle.unpack_variant(d, data, a)
                        sb = ref.unpack_variant(d, data, b)
                        assert sa == sb, (variant, big, w, h, cut, sa, sb)
                        assert (sa == 0) == (cut == 0)
                        assert np.array_equal(a.u16(), b.u16())
                        n_ok += sa == 0
                        n_err += sa != 0
    assert n_ok > 250 and n_err > 100


def test_unpack_vs_ref_sweep(oracle, ref):
    from rawspeed_amd import abi
    rng = np.random.default_rng(0)
    n = 0
    for order in range(4):
        for bps in range(1, 17):
            for w in (8, 24, 40):
                for pad in (0, 1, 3):
                    for oy in (0, 2):
                        if (w * bps) % 8:
                            continue
                        h = 5
                        pitch = w * bps // 8 + pad
                        data = rng.integers(0, 256, size=h * pitch, dtype=np.uint8)
                        d = abi.UnpackDesc(0, oy, w, h, pitch, bps, order)
                        hi, ri = HostImage(w, h + oy), ref.image(w, h + oy)
                        assert oracle.unpack(d, data, hi) == ref.unpack(d, data, ri)
                        assert np.array_equal(hi.u16(), ri.u16()), (order, bps, w, pad, oy)
                        n += 1
    assert n > 1000


def test_unpack_errors_vs_ref(oracle, ref):
    from rawspeed_amd import abi
    data = np.zeros(64, np.uint8)
    for d in [abi.UnpackDesc(0, 0, 8, 4, 12, 12, 0),     # ok
              abi.UnpackDesc(0, 0, 8, 40, 12, 12, 0),    # not enough rows -> IOE
              abi.UnpackDesc(0, 0, 0, 4, 12, 12, 0),     # empty tile
              abi.UnpackDesc(0, 0, 8, 4, 11, 12, 0),     # pitch too small
              abi.UnpackDesc(0, 0, 8, 4, 12, 12, 4),     # JPEG order
              abi.UnpackDesc(0, 0, 8, 4, 12, 17, 0),     # bps > 16
              abi.UnpackDesc(0, 0, 7, 4, 12, 12, 0),     # bits % 8
              abi.UnpackDesc(0, 5, 8, 4, 12, 12, 0),     # oy > dim.y
              abi.UnpackDesc(1, 0, 8, 4, 12, 12, 0),     # ox + w > dim.x
              abi.UnpackDesc(0, 0, 8, 1, 2, 2, 1)]:      # stream < 4 bytes
        hi, ri = HostImage(8, 4), ref.image(8, 4)
        assert oracle.unpack(d, data, hi) == ref.unpack(d, data, ri), list(bytes(d))


@pytest.mark.parametrize("c", G.LJPEG_CASES, ids=lambda c: c["name"])
def test_ljpeg_vs_ref(oracle, ref, c):
    d, data, (w, h, cpp), _ = G.build_ljpeg(c)
    hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
    assert oracle.ljpeg(d, data, hi) == ref.ljpeg(d, data, ri)
    assert np.array_equal(hi.u16(), ri.u16())


def test_ljpeg_corrupt_streams_vs_ref(oracle, ref):
    """Random corruption: status, consumed bytes and (on success) pixels agree."""
    rng = np.random.default_rng(5)
    c = next(x for x in G.LJPEG_CASES if x["name"] == "medium")
    d, data, (w, h, cpp), _ = G.build_ljpeg(c)
    n_fail = 0
    for trial in range(40):
        bad = data.copy()
        if trial % 4 == 0:
            bad = bad[:rng.integers(8, len(bad) // 2)]            # truncation
        elif trial % 4 == 1:
            bad[rng.integers(0, len(bad) - 20)] = 0xFF            # early marker
        else:
            idx = rng.integers(0, len(bad) - 20, size=3)
            bad[idx] = rng.integers(0, 256, size=3)
        hi, ri = HostImage(w, h, cpp), ref.image(w, h, cpp)
        so, sr = oracle.ljpeg(d, bad, hi), ref.ljpeg(d, bad, ri)
        assert so[0] == sr[0], (trial, so, sr, ref.last_error())
        if so[0] == 0:
            assert so[1] == sr[1]
            assert np.array_equal(hi.u16(), ri.u16())
        else:
            n_fail += 1
    assert n_fail > 0


@pytest.mark.parametrize("c", G.CR2_CASES, ids=lambda c: c["name"])
def test_cr2_vs_ref(oracle, ref, c):
    d, data, (w, h, cpp), _ = G.build_cr2(c)
    cfa = "sraw" not in c
    hi, ri = HostImage(w, h, cpp, is_cfa=cfa), ref.image(w, h, cpp, is_cfa=cfa)
    assert oracle.cr2(d, data, hi) == ref.cr2(d, data, ri)
    assert np.array_equal(hi.u16(), ri.u16())


def test_cr2_geometry_vs_ref(oracle, ref):
    """Slice / frame shapes incl. wrapped slices (frame.y != dim.y) and rejects."""
    import cases as C
    rng = np.random.default_rng(9)
    d, data, img, _ = C.make_cr2_case(rng, 40, 24, 2, (3, 16, 8))
    variants = []
    for (fw, fh, ns, sw, lw) in [(20, 24, 3, 16, 8), (20, 24, 3, 16, 6), (10, 48, 2, 20, 20),
                                 (40, 12, 1, 0, 40), (20, 24, 2, 20, 20), (20, 24, 3, 14, 12),
                                 (20, 24, 1, 0, 38), (5, 96, 4, 10, 10), (20, 24, 0, 0, 40)]:
        v = type(d).from_buffer_copy(d)
        v.frame_w, v.frame_h, v.num_slices, v.slice_width, v.last_slice_width = fw, fh, ns, sw, lw
        variants.append(v)
    n_ok = 0
    for v in variants:
        hi, ri = HostImage(40, 24, 1), ref.image(40, 24, 1)
        so, sr = oracle.cr2(v, data, hi), ref.cr2(v, data, ri)
        assert so[0] == sr[0], (v.frame_w, v.frame_h, v.num_slices, so, sr, ref.last_error())
        if so[0] == 0:
            n_ok += 1
            assert so[1] == sr[1] and np.array_equal(hi.u16(), ri.u16())
    assert n_ok >= 3
