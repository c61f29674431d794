"""Parity at the BASELINE sizes against the UNMODIFIED reference (oracle/_ref), every
frame in full: configs[1] (14-bit MSB 8192x5464), configs[2] (CR2-style 6720x4480;
plus SURVEY 8(d)'s second distribution, uniform-random 14-bit values, and a frame with
blown highlights and a black border), configs[3] (8192x5464 as 2x2 DNG tiles through
AbstractDngDecompressor::decompress; plus the overhanging 8189x5462 image and the
restart-interval variant) and a batch of configs[4] frames.  The comparison is the
rstest one -- md5 of the per-line md5s (src/utilities/rstest/rstest.cpp:131-146) --
and the arrays themselves."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench_ljpeg as B
import golden_cases as G
from oracle_lib import Ref
from rawspeed_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


@pytest.fixture(scope="module")
def ref():
    if not Ref.available():
        pytest.skip("oracle/_ref is not built here")
    return Ref()


def same(got, want):
    return G.image_hash(got) == G.image_hash(want) and np.array_equal(got, want)


def run_plan(plan, inp, out):
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, cons = plan.results()
    assert rc == 0, (rc, st)
    return cons


def route(plan, inp, out):
    """which kernels a run of the plan launches (rsx_plan_kernel_table): parity holds on
    either route, but a BASELINE shape silently demoted to the multi-kernel pipeline would
    otherwise show in the bench only"""
    plan.set_timing(True)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tab = plan.kernel_table()
    plan.set_timing(False)
    return [n for n, _ in tab[0]] if tab else []


def assert_single_pass(names, also_allowed=()):
    assert any(n.startswith("lj_fast_kernel") for n in names), names
    slow = [n for n in names if ("sync" in n or "decode" in n or "rowedge" in n)
            and n not in also_allowed]
    assert not slow, names


def test_cfg2_every_frame_vs_reference(gpu, ref):
    import bench
    cfg, frames = bench.CFG2, 3
    packed, pxs = bench.make_frames(cfg, frames, 5150)
    w, h, bps = cfg["w"], cfg["h"], cfg["bps"]
    inp = torch.from_numpy(packed).cuda()
    out = torch.empty(frames * h * bench.out_pitch(w), dtype=torch.uint8, device="cuda")
    run_plan(gpu.unpack_plan(bench.unpack_jobs(cfg, frames)), inp, out)
    d = abi.UnpackDesc(0, 0, w, h, w * bps // 8, bps, cfg["order"])
    strip = h * (w * bps // 8)
    for f in range(frames):
        img = ref.image(w, h, 1)
        assert ref.unpack(d, packed[f * strip:(f + 1) * strip], img) == 0
        assert same(bench.frame_of(out, cfg, f), img.pixels()), f


def _cr2_frames(gpu, ref, made, W, H, single_pass=True):
    plan, inp, out = B._cr2_batch(gpu, torch, [(m[0], m[1]) for m in made], W, H)
    cons = run_plan(plan, inp, out)
    if single_pass:
        assert_single_pass(route(plan, inp, out))
    for f, m in enumerate(made):
        img = ref.image(W, H, 1)
        st, rcons = ref.cr2(m[0], m[1], img)
        assert st == 0 and cons[f] == rcons == m[3]
        got = B.gpu_frame(out, f, W, H)
        assert same(got, img.pixels()), f
        assert np.array_equal(got, m[2]), f


def test_cfg3_every_frame_vs_reference(gpu, ref):
    W, H = 6720, 4480
    made = [B.make_cr2_frame(W, H, (3, 2240, 2240), seed=11 + f) for f in range(2)]
    _cr2_frames(gpu, ref, made, W, H)


def _cr2_from_image(src, W, H):
    import cases
    rows = cases.cr2_stream_from_image(src, 2, W // 2, H, cases.cr2_slices(3, 2240, 2240))
    scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 3, 2240, 2240
    abi.fill_recipe(d, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(30, np.uint8)])
    return d, data, src, len(scan)


def test_cfg3_uniform_random_14bit_vs_reference(gpu, ref):
    """~20 bit/px: the entropy-coded input is larger than the output"""
    W, H = 6720, 4480
    src = synth.uniform(W * H, 14, 99).reshape(H, W)
    _cr2_frames(gpu, ref, [_cr2_from_image(src, W, H)], W, H)


def test_cfg3_blown_highlights_vs_reference(gpu, ref):
    """10 % saturated at 16383 + a black border: constant regions = periodic bit stream"""
    W, H = 6720, 4480
    _cr2_frames(gpu, ref, [_cr2_from_image(B.clipped_image(W, H, 55), W, H)], W, H)


@pytest.mark.parametrize("shape", [(8192, 5464, 0), (8189, 5462, 0), (8192, 5464, 28)],
                         ids=["cfg4", "overhang_8189x5462", "restart_intervals"])
def test_cfg4_dng_tiles_vs_reference_fanout(gpu, ref, shape):
    W, H, ri = shape
    tw, th = 4096, 2732
    src, jobs, datas, blobs, lens = B._dng_tiles(W, H, tw, th, 21, ri)
    inp = torch.from_numpy(np.concatenate(datas)).cuda()
    out = torch.zeros(B.out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = gpu.ljpeg_plan(jobs)
    cons = run_plan(plan, inp, out)
    if ri == 0:  # (restart intervals: the intervals are streams of a child plan)
        assert_single_pass(route(plan, inp, out))
    img = ref.image(W, H, 1)
    assert ref.dng(img, 7, tw, th, blobs, threads=4) == 0, ref.last_error()
    got = B.gpu_frame(out, 0, W, H)
    assert same(got, img.pixels())
    assert np.array_equal(got, src)
    if H % th == 0:
        assert cons == lens
    # consumed of every tile against LJpegDecompressor::decode on the same scan
    for j, data, c in zip(jobs, datas, cons):
        timg = ref.image(W, H, 1)
        st, rc = ref.ljpeg(j.desc, data, timg)
        assert st == 0 and rc == c


def test_cfg5_batch_every_frame_vs_reference(gpu, ref):
    frames = 6
    plan, inp, out, meta = B.make_cfg5_plan(gpu, torch, frames, distinct=3, seed0=4000)
    cons = run_plan(plan, inp, out)
    assert_single_pass(route(plan, inp, out))
    refs = []
    for d, data in meta["blobs"]:
        img = ref.image(meta["W"], meta["H"], 1)
        st, rc = ref.ljpeg(d, data, img)
        assert st == 0
        refs.append(img.pixels().copy())
    assert B.check_cfg5(out, meta, cons, frames, refs)
    for f in range(frames):
        assert same(B.gpu_frame(out, f, meta["W"], meta["H"]), refs[meta["pick"][f]]), f
