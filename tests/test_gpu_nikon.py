"""GPU parity: NikonDecompressor through the C-ABI (rsx_nikon_decompress /
rsx_nikon_plan_create) vs the oracle and the reference's golden hashes."""
import json
import os

import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import golden_cases as G
import nikon_cases as N
from oracle_lib import HostImage, out_pitch

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "golden_hashes.json")) as f:
    GOLD = json.load(f)


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


@pytest.mark.parametrize("c", G.NIKON_CASES, ids=lambda c: c["name"])
def test_nikon_golden(gpu, oracle, c):
    meta, d, data, (w, h, cpp), src = G.build_nikon(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st = gpu.nikon_decompress(d, data, img.view())
    assert st == oracle.nikon(d, data, want) == GOLD["nikon"][c["name"]]["status"] == 0
    assert np.array_equal(img.u16(), want.u16())
    assert G.image_hash(img.pixels()) == GOLD["nikon"][c["name"]]["hash"]
    if src is not None and c["unc"]:
        assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("bits,w,h,unc", [(14, 2144, 400, 1), (14, 2144, 400, 0),
                                          (12, 1000, 333, 0), (14, 8288, 64, 0)])
def test_nikon_lossless_sizes(gpu, oracle, bits, w, h, unc):
    """Several workgroups per stream, odd heights, the maximum width."""
    rng = np.random.default_rng([51, bits, w])
    src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1)
    meta = N.metadata(70, 0, [2000, 2100, 2200, 2300])
    P = N.parse(meta, bits, h)
    pu = P["p_up"]
    data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                 synth.NIKON_TREE[P["huff_select"]])
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    d = N.desc(P, bits, bool(unc))
    img, want = HostImage(w, h), HostImage(w, h)
    assert gpu.nikon_decompress(d, data, img.view()) == oracle.nikon(d, data, want) == 0
    assert np.array_equal(img.u16(), want.u16())
    if unc:
        assert np.array_equal(img.pixels(), src)


@pytest.mark.parametrize("bits,v1,w,h,split", [(12, 32, 1504, 300, 131), (14, 32, 1000, 240, 1),
                                               (14, 64, 640, 200, 199), (12, 32, 2000, 90, 45)])
def test_nikon_split_sizes(gpu, oracle, bits, v1, w, h, split):
    """The "lossy after split" table starts at a bit position that only the
    decode of the first part reveals (two-phase plan); clamping at 0 / 32767."""
    rng = np.random.default_rng([52, bits, w, split])
    pts = G.nikon_curve_points(257, (1 << (bits - 2 if v1 == 64 else bits)) - 1)
    meta = N.metadata(68, v1, [3000, 3100, 3200, 3300], pts, split, pad_to=3000)
    P = N.parse(meta, bits, h)
    assert P["split"] == split
    hs = P["huff_select"]
    data = N.symbol_stream(rng, split * w, synth.NIKON_TREE[hs], (h - split) * w,
                           synth.NIKON_TREE[hs + 1])
    for unc in (1, 0):
        d = N.desc(P, bits, bool(unc))
        img, want = HostImage(w, h), HostImage(w, h)
        assert gpu.nikon_decompress(d, data, img.view()) == oracle.nikon(d, data, want) == 0
        assert np.array_equal(img.u16(), want.u16())


def test_nikon_curve_mutated_in_place_between_calls(gpu, oracle):
    """The lane keeps the plan of its last call, keyed by what the plan was made FROM.  The
    curve is handed over as a host pointer and baked into the plan (dither table): two
    calls with the same descriptor and the same curve ADDRESS but other curve CONTENTS must
    not share a plan (the key holds the contents, rsx_api.hip key_job)."""
    bits, w, h = 12, 640, 96
    rng = np.random.default_rng(77)
    pts = G.nikon_curve_points(300, 4000)
    meta = N.metadata(68, 0, [2000, 2100, 2200, 2300], pts)
    P = N.parse(meta, bits, h)
    data = N.symbol_stream(rng, w * h, synth.NIKON_TREE[P["huff_select"]])
    d = N.desc(P, bits, False)
    addr = d.curve
    outs = []
    for round_ in range(3):
        img, want = HostImage(w, h), HostImage(w, h)
        assert gpu.nikon_decompress(d, data, img.view()) == oracle.nikon(d, data, want) == 0
        assert np.array_equal(img.u16(), want.u16()), round_
        outs.append(img.u16().copy())
        # same buffer, other contents (still monotonic, still the same size)
        d._curve[:] = np.minimum(d._curve.astype(np.uint32) * 3 // 4 + 5 * (round_ + 1),
                                 65535).astype(np.uint16)
        assert d.curve == addr
    assert not np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[1], outs[2])


def test_nikon_truncated(gpu, oracle):
    """Status parity at every cut of a split stream (BitStreamerMSB reads zeros
    for 8 bytes past the end, then throws)."""
    c = dict(name="trunc", v0=68, v1=32, bits=12, w=40, h=16, split=7, kind="symbols", unc=1)
    meta, d, data, (w, h, cpp), _ = G.build_nikon(c)
    full = int(np.flatnonzero(data)[-1]) + 1
    seen = set()
    for cut in list(range(0, 40)) + [60, 100, 200, full // 2, full - 3]:
        n = full - cut
        if n < 1:
            continue
        part = data[:n]
        img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
        so = oracle.nikon(d, part, want)
        sg = gpu.nikon_decompress(d, part, img.view())
        assert sg == so, (cut, sg, so)
        if so == 0:
            assert np.array_equal(img.u16(), want.u16())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2


def test_nikon_plan_batch(gpu, oracle):
    """Device-resident plan: several frames (with and without split, both output
    modes) in one launch sequence; run twice (the split child plan is reused)."""
    import gpu_util
    rng = np.random.default_rng(53)
    jobs, wants, chunks = [], [], []
    in_off = out_off = 0
    keep = []
    for k, (bits, split, unc) in enumerate([(14, 0, 1), (12, 40, 0), (14, 0, 0), (12, 17, 1)]):
        w, h = 640 + 64 * k, 120
        if split:
            pts = G.nikon_curve_points(257, (1 << bits) - 1)
            meta = N.metadata(68, 32, [3000, 3100, 3200, 3300], pts, split, pad_to=3000)
            P = N.parse(meta, bits, h)
            hs = P["huff_select"]
            data = N.symbol_stream(rng, split * w, synth.NIKON_TREE[hs], (h - split) * w,
                                   synth.NIKON_TREE[hs + 1])
        else:
            meta = N.metadata(70, 0, [2000, 2100, 2200, 2300])
            P = N.parse(meta, bits, h)
            src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1)
            pu = P["p_up"]
            data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                         synth.NIKON_TREE[P["huff_select"]])
            data = np.concatenate([data, np.zeros(8, np.uint8)])
        d = N.desc(P, bits, bool(unc))
        keep.append(d)
        want = HostImage(w, h)
        assert oracle.nikon(d, data, want) == 0
        j = abi.NikonJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = in_off, data.size, out_off
        j.img = gpu_util.image_job_view(w, h, 1, want.pitch)
        jobs.append(j)
        wants.append(want)
        chunks.append((in_off, data))
        in_off += (data.size + 64 + 15) // 16 * 16
        out_off += want.buf.size
    in_host = np.zeros(in_off + 64, np.uint8)
    for off, data in chunks:
        in_host[off:off + data.size] = data
    d_in = gpu_util.to_dev(in_host)
    plan = gpu.nikon_plan(jobs)
    for _ in range(2):
        d_out = torch.full((out_off + 16,), 0xA5, dtype=torch.uint8, device="cuda")
        plan.run(d_in.data_ptr(), d_out.data_ptr())
        rc, status, _ = plan.results()
        assert rc == 0 and status == [0] * len(jobs)
        got = d_out.cpu().numpy()
        for j, want in zip(jobs, wants):
            assert np.array_equal(got[j.img_offset:j.img_offset + want.buf.size], want.buf)
    plan.close()


# ---- PentaxDecompressor ----------------------------------------------------------

@pytest.mark.parametrize("c", G.PENTAX_CASES, ids=lambda c: c["name"])
def test_pentax_golden(gpu, oracle, c):
    meta, d, data, (w, h, cpp), src = G.build_pentax(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st = gpu.pentax_decompress(d, data, img.view())
    assert st == oracle.pentax(d, data, want)
    if st == 0:
        assert np.array_equal(img.u16(), want.u16())
        assert G.image_hash(img.pixels()) == GOLD["pentax"][c["name"]]["hash"]
        assert np.array_equal(img.pixels(), src)
    else:
        assert st == abi.RSX_ERR_VALUE_RANGE


def test_pentax_sizes_and_truncation(gpu, oracle):
    rng = np.random.default_rng(54)
    for tree, w, h in ((synth.PENTAX_TREE, 4000, 333), (N.PENTAX_MODERN, 6000, 120)):
        src = N.smooth15(rng, h, w, maxv=16383 if tree is N.PENTAX_MODERN else 4095, sigma=6.0)
        data, _ = N.pentax_encode(src, tree)
        d = N.pentax_desc(tree)
        full = np.concatenate([data, np.zeros(8, np.uint8)])
        img, want = HostImage(w, h), HostImage(w, h)
        assert gpu.pentax_decompress(d, full, img.view()) == oracle.pentax(d, full, want) == 0
        assert np.array_equal(img.u16(), want.u16()) and np.array_equal(img.pixels(), src)
        for cut in (1, 5, 9, 13, 40, len(data) // 2):
            part = data[:len(data) - cut]
            img, want = HostImage(w, h), HostImage(w, h)
            so = oracle.pentax(d, part, want)
            assert gpu.pentax_decompress(d, part, img.view()) == so, cut
            if so == 0:
                assert np.array_equal(img.u16(), want.u16())


# ---- SamsungV1Decompressor -------------------------------------------------------

@pytest.mark.parametrize("c", G.SAMSUNG_V1_CASES, ids=lambda c: c["name"])
def test_samsung_v1_golden(gpu, oracle, c):
    d, data, (w, h, cpp), src = G.build_samsung_v1(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st = gpu.samsung_v1_decompress(d, data, img.view())
    assert st == oracle.samsung_v1(d, data, want)
    if st == 0:
        assert np.array_equal(img.u16(), want.u16())
        assert G.image_hash(img.pixels()) == GOLD["samsung_v1"][c["name"]]["hash"]
        assert np.array_equal(img.pixels(), src)
    else:
        assert st == abi.RSX_ERR_VALUE_RANGE


def test_samsung_v1_truncation(gpu, oracle):
    """fill(23) instead of fill(32): symbols may start 9 bits later before the
    bit streamer overflows -- status parity at every cut."""
    rng = np.random.default_rng(55)
    w, h = 2048, 64
    src = N.smooth15(rng, h, w, maxv=4095, sigma=6.0)
    data, _ = synth.prefix_encode(src, [0, 0, 0, 0], synth.SAMSUNG_V1_TAB)
    d = abi.SamsungV1Desc.make(synth.SAMSUNG_V1_TAB)
    seen = set()
    for cut in range(0, 24):
        part = data[:len(data) - cut]
        img, want = HostImage(w, h), HostImage(w, h)
        so = oracle.samsung_v1(d, part, want)
        assert gpu.samsung_v1_decompress(d, part, img.view()) == so, cut
        if so == 0:
            assert np.array_equal(img.u16(), want.u16())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2


# ---- SonyArw1Decompressor ----------------------------------------------------------

@pytest.mark.parametrize("c", G.SONY_ARW1_CASES, ids=lambda c: c["name"])
def test_sony_arw1_golden(gpu, oracle, c):
    data, (w, h, cpp), src = G.build_sony_arw1(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    st = gpu.sony_arw1_decompress(data, img.view())
    assert st == oracle.sony_arw1(data, want)
    if st == 0:
        assert np.array_equal(img.u16(), want.u16())
        assert G.image_hash(img.pixels()) == GOLD["sony_arw1"][c["name"]]["hash"]
        assert np.array_equal(img.pixels(), src)
    else:
        assert st == abi.RSX_ERR_VALUE_RANGE


def test_sony_arw1_full_frame_and_truncation(gpu, oracle):
    """The A100's 3881 x 2608 frame (ArwDecoder.cpp:128-129: odd width, several
    hundred workgroups in one stream), then status parity at every cut."""
    rng = np.random.default_rng(57)
    w, h = 3881, 2608
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    src = np.clip(600 + 0.5 * x + 0.4 * y + rng.normal(0, 9, (h, w)), 0, 4095).astype(np.uint16)
    data, _ = synth.sony_arw1_encode(src)
    img, want = HostImage(w, h), HostImage(w, h)
    assert gpu.sony_arw1_decompress(data, img.view()) == oracle.sony_arw1(data, want) == 0
    assert np.array_equal(img.u16(), want.u16()) and np.array_equal(img.pixels(), src)
    w, h = 300, 64
    data, _ = synth.sony_arw1_encode(src[:h, :w])
    seen = set()
    for cut in list(range(0, 24)) + [len(data) // 2]:
        part = data[:len(data) - cut]
        img, want = HostImage(w, h), HostImage(w, h)
        so = oracle.sony_arw1(part, want)
        assert gpu.sony_arw1_decompress(part, img.view()) == so, cut
        if so == 0:
            assert np.array_equal(img.u16(), want.u16())
        seen.add(so)
    assert 0 in seen and len(seen) >= 2


def test_sony_arw1_damaged(gpu, oracle):
    """Random damage: whenever the reference algorithm succeeds, same pixels;
    otherwise a failure (which error comes first is not part of the contract)."""
    rng = np.random.default_rng(58)
    w, h = 257, 128
    src = np.clip(1500 + rng.normal(0, 40, (h, w)).cumsum(axis=0) * 0.2, 0, 4095).astype(np.uint16)
    data, _ = synth.sony_arw1_encode(src)
    n_ok = n_bad = 0
    for trial in range(16):
        bad = data.copy()
        idx = rng.integers(0, len(bad), size=2)
        bad[idx] ^= np.uint8(1) << rng.integers(0, 8, size=2).astype(np.uint8)
        img, want = HostImage(w, h), HostImage(w, h)
        so = oracle.sony_arw1(bad, want)
        sg = gpu.sony_arw1_decompress(bad, img.view())
        if so == 0:
            n_ok += 1
            assert sg == 0 and np.array_equal(img.u16(), want.u16()), trial
        else:
            n_bad += 1
            assert sg != 0, trial
    assert n_bad > 0


# ---- HasselbladDecompressor -------------------------------------------------------

@pytest.mark.parametrize("c", G.HASSELBLAD_CASES, ids=lambda c: c["name"])
def test_hasselblad_golden(gpu, oracle, c):
    d, data, (w, h, cpp), src = G.build_hasselblad(c)
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    g = GOLD["hasselblad"][c["name"]]
    sg = gpu.hasselblad_decompress(d, data, img.view())
    assert sg == oracle.hasselblad(d, data, want) == (0, g["consumed"])
    assert np.array_equal(img.u16(), want.u16())
    assert G.image_hash(img.pixels()) == g["hash"]
    assert np.array_equal(img.pixels(), src)


def test_hasselblad_sizes_and_truncation(gpu, oracle):
    """Several workgroups per stream; status and getStreamPosition() parity at every
    cut (partial last MSB32 word, 8-byte position budget)."""
    import cases as C
    rng = np.random.default_rng(56)
    for w, h, full in ((4000, 250, False), (2048, 96, True)):
        src = (rng.integers(0, 65536, size=(h, w), dtype=np.uint16) if full
               else C.smooth_image(rng, h, w, 14))
        data, _ = synth.hasselblad_encode(src, 0x4000, C.FULL17)
        d = abi.HasselbladDesc.make(C.FULL17, 0x4000)
        img, want = HostImage(w, h), HostImage(w, h)
        so = oracle.hasselblad(d, data, want)
        assert so[0] == 0 and gpu.hasselblad_decompress(d, data, img.view()) == so
        assert np.array_equal(img.u16(), want.u16()) and np.array_equal(img.pixels(), src)
        seen = set()
        for cut in list(range(1, 20)) + [40, 101, len(data) // 2]:
            part = data[:len(data) - cut]
            img, want = HostImage(w, h), HostImage(w, h)
            so = oracle.hasselblad(d, part, want)
            sg = gpu.hasselblad_decompress(d, part, img.view())
            assert sg[0] == so[0], (cut, sg, so)
            if so[0] == 0:
                assert sg == so and np.array_equal(img.u16(), want.u16())
            seen.add(so[0])
        assert len(seen) >= 2
