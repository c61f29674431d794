"""GPU parity of the three routes a Nikon-type stream can take (round 6):

  pixels by the single-pass kernel      lj_fast_kernel<nikon-type>   (fast_nk)
  differences by the single-pass kernel lj_fast_kernel<differences>  (fast_diffs) + K5 / K6
  the legacy route                      lj_decode_kernel ...          + K5 / K6

The first one's sums are mod 2^16 where the reference's are ints (NikonDecompressor.cpp:518-560,
PentaxDecompressor.cpp:155-177): a value outside 0 .. 32767 makes the kernel hand the stream to
the legacy route.  RSX_NO_FAST_NK / RSX_NO_FAST_DIFFS in the environment (read when a plan is
made) take the route one further back; every route must give the oracle's image and status.
"""
import os

import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import golden_cases as G
import nikon_cases as N
from oracle_lib import HostImage

pytestmark = pytest.mark.gpu

ROUTES = ((), ("RSX_NO_FAST_NK",), ("RSX_NO_FAST_NK", "RSX_NO_FAST_DIFFS"))
# RSX_FUZZ_BASE=<k> moves the random-shape cases to other seeds (soak runs)
BASE = int(os.environ.get("RSX_FUZZ_BASE", "0"))


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _encode_ints(src, p_up, tree):
    """NikonDecompressor's stream (no split) of an image given as INTS -- values below 0 or above
    32767 are what the decoder's sums reach before clampBits(., 15) -- by the predictor of
    NikonDecompressor.cpp:518-560: the first pair of a row from the row two above, the others from
    the sample two to the left."""
    h, w = src.shape
    pred = np.empty_like(src)
    pred[:, 2:] = src[:, :-2]
    up = np.array(p_up, np.int64).reshape(2, 2)
    for y in range(h):
        pred[y, :2] = up[y & 1]
        up[y & 1] = src[y, :2]
    diff = (src - pred).ravel()
    mag = np.abs(diff)
    ssss = np.where(mag == 0, 0, np.floor(np.log2(np.maximum(mag, 1))).astype(np.int64) + 1)
    assert ssss.max() <= 15
    by_len = {v: (c, l) for (c, l, v) in N._canonical(tree)}
    code = np.array([by_len[int(v)][0] for v in ssss], np.int64)
    clen = np.array([by_len[int(v)][1] for v in ssss], np.int64)
    extra = np.where(diff >= 0, diff, diff + (1 << ssss) - 1)
    val, ln = (code << ssss) | extra, clen + ssss
    start = np.concatenate([[0], np.cumsum(ln)[:-1]])
    bits = np.zeros(int(ln.sum()) + 8, np.uint8)
    for k in range(int(ln.max())):
        m = k < ln
        bits[start[m] + k] = (val[m] >> (ln[m] - 1 - k)) & 1
    return np.concatenate([np.packbits(bits), np.zeros(16, np.uint8)])


class _env:
    def __init__(self, names):
        self.names = names

    def __enter__(self):
        for n in self.names:
            os.environ[n] = "1"

    def __exit__(self, *a):
        for n in self.names:
            os.environ.pop(n, None)


def _run(gpu, make_plan, jobs, in_host, out_bytes, route):
    import gpu_util
    with _env(route):
        plan = make_plan(jobs)
    d_in = gpu_util.to_dev(in_host)
    d_out = torch.full((out_bytes + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    plan.set_timing(True)
    plan.run(d_in.data_ptr(), d_out.data_ptr())
    rc, status, _ = plan.results()
    tab = plan.kernel_table()
    plan.kernel_time()  # (resets the totals: the next table is the next run's)
    names = [n for n, _ in tab[0]] if tab else []
    # a second run of the same plan: the steady-state instantiation, the cached level -- and the pass
    # that redoes what the single-pass kernel gave up on is launched by the run itself (the first
    # run's is launched when the results are fetched, outside the timed launches): its kernels are
    # in THIS run's table
    d_out2 = torch.full((out_bytes + 16,), 0xA5, dtype=torch.uint8, device="cuda")
    plan.run(d_in.data_ptr(), d_out2.data_ptr())
    rc2, status2, _ = plan.results()
    tab = plan.kernel_table()
    names += ["run 2: " + n for n, _ in tab[0]] if tab else []
    plan.set_timing(False)
    plan.close()
    assert rc == rc2 and status == status2  # (rc: the first failing job's status)
    a, b = d_out.cpu().numpy(), d_out2.cpu().numpy()
    return status, a, b, names


def _nikon_job(gpu_util, d, data, w, h, pitch, in_off, out_off):
    j = abi.NikonJob()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = in_off, data.size, out_off
    j.img = gpu_util.image_job_view(w, h, 1, pitch)
    return j


@pytest.mark.parametrize("bits,w,h,unc,kind", [
    (14, 2144, 300, 1, "image"), (14, 2144, 300, 0, "image"), (12, 1000, 333, 0, "image"),
    (14, 6016, 40, 0, "image"), (14, 1504, 200, 1, "walk"), (12, 640, 240, 0, "walk"),
    (14, 30, 57, 0, "image"), (14, 2, 9, 0, "image")])
def test_nikon_routes_agree(gpu, oracle, bits, w, h, unc, kind):
    """"image": a smooth image (every value inside the sensor's bits: the single-pass kernel writes
    the pixels); "walk": random symbols, the sums wander below 0 and above 32767 (clampBits at
    either end -- the kernel must notice and the legacy route redo the stream)."""
    import gpu_util
    rng = np.random.default_rng([61, bits, w, h, unc])
    meta = N.metadata(70, 0, [2000, 2100, 2200, 2300]) if kind == "image" else \
        N.metadata(68, 0, [3000, 3100, 3200, 3300], G.nikon_curve_points(257, (1 << bits) - 1))
    P = N.parse(meta, bits, h)
    if kind == "image":
        src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1)
        pu = P["p_up"]
        data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                     synth.NIKON_TREE[P["huff_select"]])
        data = np.concatenate([data, np.zeros(8, np.uint8)])
    else:
        data = N.symbol_stream(rng, w * h, synth.NIKON_TREE[P["huff_select"]])
    d = N.desc(P, bits, bool(unc))
    # three frames in a plan: the same stream at three offsets of the output
    want = HostImage(w, h)
    assert oracle.nikon(d, data, want) == 0
    stride_in = (data.size + 64 + 15) // 16 * 16
    in_host = np.zeros(3 * stride_in + 64, np.uint8)
    jobs = []
    for k in range(3):
        in_host[k * stride_in:k * stride_in + data.size] = data
        jobs.append(_nikon_job(gpu_util, d, data, w, h, want.pitch, k * stride_in, k * want.buf.size))
    clamped = kind == "walk" and bool((want.u16()[:, :w] == 0).any() or (want.u16()[:, :w] >= 32767).any())
    for route in ROUTES:
        status, a, b, names = _run(gpu, gpu.nikon_plan, jobs, in_host, 3 * want.buf.size, route)
        assert status == [0, 0, 0], (route, status)
        for k in range(3):
            for got in (a, b):
                assert np.array_equal(got[k * want.buf.size:(k + 1) * want.buf.size], want.buf), (route, k)
        if route == ():
            assert any("nikon-type" in x for x in names), names
            redone = bool([x for x in names if "legacy" in x or "decode" in x])
            assert redone == (kind == "walk"), names
        elif route == ("RSX_NO_FAST_NK",):
            assert any("differences" in x for x in names) and not any("nikon-type" in x for x in names), names
        else:
            assert not any("lj_fast_kernel" in x for x in names), names
    if kind == "walk":
        assert clamped  # (the case is about the clamp: a seed that never reaches it tests nothing)


def test_nikon_values_at_the_edges(gpu, oracle):
    """Values 0 and 32767 are inside (nothing redone); a stream whose only excursion is ONE sample
    at 32768 or at -1 is redone and comes out clamped."""
    import gpu_util
    bits, w, h = 14, 256, 64
    meta = N.metadata(70, 0, [2000, 2100, 2200, 2300])
    P = N.parse(meta, bits, h)
    pu = P["p_up"]
    tree = synth.NIKON_TREE[P["huff_select"]]
    rng = np.random.default_rng(62)
    base = N.smooth15(rng, h, w, maxv=16383).astype(np.int64)
    for name, (y, x, val), redone in (("inside", (20, 100, 32767), False), ("inside0", (21, 7, 0), False),
                                      ("above", (33, 201, 32768), True), ("below", (5, 0, -1), True)):
        src = base.copy()
        src[y, x] = val
        if val > 16383:  # (the tree's largest difference has 14 bits: a ramp of same-colour neighbours)
            src[y, x - 4] = src[y, x + 4] = 16000
            src[y, x - 2] = src[y, x + 2] = 24000
        data = _encode_ints(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]], tree)
        d = N.desc(P, bits, True)
        want = HostImage(w, h)
        assert oracle.nikon(d, data, want) == 0
        assert np.array_equal(want.u16()[:, :w], np.clip(src, 0, 32767).astype(np.uint16)), name
        jobs = [_nikon_job(gpu_util, d, data, w, h, want.pitch, 0, 0)]
        in_host = np.concatenate([data, np.zeros(64, np.uint8)])
        status, a, b, names = _run(gpu, gpu.nikon_plan, jobs, in_host, want.buf.size, ())
        assert status == [0]
        assert np.array_equal(a[:want.buf.size], want.buf) and np.array_equal(b[:want.buf.size], want.buf), name
        assert any("nikon-type" in x for x in names), names
        assert bool([x for x in names if "legacy" in x]) == redone, (name, names)


def test_pentax_routes_agree(gpu, oracle):
    """Pentax on the three routes: an image inside the sensor's bits, and one with a value that does
    not fit 16 bits (an error of the reference's, PentaxDecompressor.cpp:170: the status must be the
    legacy route's on every route)."""
    import gpu_util
    rng = np.random.default_rng(63)
    for tree, w, h, maxv in ((synth.PENTAX_TREE, 4000, 120, 4095), (N.PENTAX_MODERN, 6000, 64, 16383)):
        src = N.smooth15(rng, h, w, maxv=maxv, sigma=6.0)
        data, _ = N.pentax_encode(src, tree)
        d = N.pentax_desc(tree)
        full = np.concatenate([data, np.zeros(8, np.uint8)])
        want = HostImage(w, h)
        assert oracle.pentax(d, full, want) == 0
        j = abi.PentaxJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = 0, full.size, 0
        j.img = gpu_util.image_job_view(w, h, 1, want.pitch)
        in_host = np.concatenate([full, np.zeros(64, np.uint8)])
        for route in ROUTES:
            status, a, b, names = _run(gpu, gpu.pentax_plan, [j], in_host, want.buf.size, route)
            assert status == [0]
            assert np.array_equal(a[:want.buf.size], want.buf) and np.array_equal(b[:want.buf.size], want.buf)
            if route == ():
                assert any("nikon-type" in x for x in names) and not [x for x in names if "legacy" in x], names
    for c in G.PENTAX_CASES:
        meta, d, data, (w, h, cpp), src = G.build_pentax(c)
        want = HostImage(w, h, cpp)
        so = oracle.pentax(d, data, want)
        j = abi.PentaxJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
        j.img = gpu_util.image_job_view(w, h, cpp, want.pitch)
        in_host = np.concatenate([data, np.zeros(64, np.uint8)])
        for route in ROUTES:
            status, a, b, names = _run(gpu, gpu.pentax_plan, [j], in_host, want.buf.size, route)
            assert status == [so], (c["name"], route, status, so)
            if so == 0:
                assert np.array_equal(a[:want.buf.size], want.buf), (c["name"], route)


@pytest.mark.parametrize("seed", range(8))
def test_nikon_type_plans_of_random_shapes(gpu, oracle, seed):
    """Plans of two to four Nikon frames of unrelated sizes (widths from 2 samples, single rows, rows
    that do not start on the 16-byte grid of the batch buffer), both output modes, curves of
    several sizes, predictors anywhere inside the sensor's bits: against the oracle, twice."""
    import gpu_util
    rng = np.random.default_rng([64, BASE, seed])
    jobs, wants, chunks, keep = [], [], [], []
    in_off = out_off = 0
    for k in range(int(rng.integers(2, 5))):
        bits = int(rng.choice([12, 14]))
        w = 2 * int(rng.integers(1, 1600)) if rng.random() < 0.8 else 2 * int(rng.integers(1, 9))
        h = int(rng.integers(1, 120))
        unc = bool(rng.integers(0, 2))
        pu = [int(v) for v in rng.integers(0, 1 << bits, size=4)]
        pts = None if rng.random() < 0.4 else G.nikon_curve_points(int(rng.choice([33, 257, 300])),
                                                                   (1 << bits) - 1)
        meta = N.metadata(70, 0, pu) if pts is None else N.metadata(68, 0, pu, pts, pad_to=3000)
        P = N.parse(meta, bits, h)
        src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1, sigma=float(rng.choice([2.0, 12.0, 60.0])))
        p = P["p_up"]
        try:
            data, _ = synth.nikon_encode(src, [p[0][0], p[0][1], p[1][0], p[1][1]],
                                         synth.NIKON_TREE[P["huff_select"]])
        except ValueError:  # (a difference the tree has no code for: a flatter image)
            src = N.smooth15(rng, h, w, maxv=(1 << (bits - 1)) - 1, sigma=2.0)
            data, _ = synth.nikon_encode(src, [p[0][0], p[0][1], p[1][0], p[1][1]],
                                         synth.NIKON_TREE[P["huff_select"]])
        data = np.concatenate([data, np.zeros(8, np.uint8)])
        d = N.desc(P, bits, unc)
        keep.append(d)
        want = HostImage(w, h)
        assert oracle.nikon(d, data, want) == 0
        jobs.append(_nikon_job(gpu_util, d, data, w, h, want.pitch, in_off, out_off))
        wants.append(want)
        chunks.append((in_off, data))
        in_off += data.size + int(rng.integers(0, 40))  # (streams at any byte offset)
        out_off += want.buf.size
    in_host = np.zeros(in_off + 64, np.uint8)
    for off, data in chunks:
        in_host[off:off + data.size] = data
    status, a, b, names = _run(gpu, gpu.nikon_plan, jobs, in_host, out_off, ())
    assert status == [0] * len(jobs)
    for j, want in zip(jobs, wants):
        for got in (a, b):
            assert np.array_equal(got[j.img_offset:j.img_offset + want.buf.size], want.buf), names
    assert any("nikon-type" in x for x in names), names


@pytest.mark.parametrize("w,h", [(2, 700), (4, 1500), (6, 300)])
def test_nikon_more_rows_than_a_workgroup_takes(gpu, oracle, w, h):
    """A workgroup of the single-pass kernel takes at most 256 stream rows (LF_RMAX); a frame a few
    pixels wide has more of them in a workgroup's bytes: the kernel hands such a stream over
    (whatever route ends up decoding it, the image is the oracle's)."""
    import gpu_util
    bits = 14
    rng = np.random.default_rng([65, w, h])
    meta = N.metadata(70, 0, [2000, 2100, 2200, 2300])
    P = N.parse(meta, bits, h)
    src = N.smooth15(rng, h, w, maxv=(1 << bits) - 1, sigma=3.0)
    pu = P["p_up"]
    data, _ = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]], synth.NIKON_TREE[P["huff_select"]])
    data = np.concatenate([data, np.zeros(8, np.uint8)])
    for unc in (True, False):
        d = N.desc(P, bits, unc)
        want = HostImage(w, h)
        assert oracle.nikon(d, data, want) == 0
        jobs = [_nikon_job(gpu_util, d, data, w, h, want.pitch, 0, 0)]
        in_host = np.concatenate([data, np.zeros(64, np.uint8)])
        status, a, b, names = _run(gpu, gpu.nikon_plan, jobs, in_host, want.buf.size, ())
        assert status == [0]
        assert np.array_equal(a[:want.buf.size], want.buf) and np.array_equal(b[:want.buf.size], want.buf), names
        img = HostImage(w, h)
        assert gpu.nikon_decompress(d, data, img.view()) == 0
        assert np.array_equal(img.buf, want.buf)
