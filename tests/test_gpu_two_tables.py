"""Two Huffman tables that alternate symbol by symbol -- one per component of a two-component
scan, what DNG writers emit; A B A B over four components -- on the single-pass kernel
(round 4: LjStreamDev::fast == 2, lj_fast_kernel<N, true>).

The reference binds one decoder per component (AbstractLJpegDecoder.h:112-125,
LJpegDecompressor.cpp:184-251: `ht[i]` of component i inside the MCU loop), so the table
of a symbol is its index in the stream mod N.  For the kernels the table of the NEXT
symbol is part of every parse state (start guesses, chain links, look-back records).
Everything here is checked against the oracle, bit by bit, and the route is asserted from
the kernels that ran."""
import numpy as np
import pytest
import torch

import cases as C
from rawspeed_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _kernel_names(plan, inp, out):
    s = torch.cuda.current_stream().cuda_stream
    plan.set_timing(True)
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    tab = plan.kernel_table()
    plan.set_timing(False)
    return [n for n, _ in tab[0]] if tab else []


def _run(gpu, oracle, d, data, W, H, expect_fast=True, cpp=1):
    import bench_ljpeg as B
    from oracle_lib import HostImage
    want = HostImage(W, H, cpp=cpp) if cpp != 1 else HostImage(W, H)
    st_o, cons_o = oracle.ljpeg(d, data, want)
    assert st_o == 0
    j = abi.LJpegJob()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    op = B.out_pitch(W * cpp)
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, cpp, int(cpp == 1)
    plan = gpu.ljpeg_plan([j])
    inp = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):  # (the second run: adaptive launches, a demoted stream)
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and not any(st) and list(cons) == [cons_o]
        px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W * cpp]
        assert np.array_equal(px, want.pixels())
    return plan, inp, out


@pytest.mark.parametrize("mcu,other", [((2, 1), "alt"), ((2, 1), "random"), ((4, 1), "alt"),
                                       ((4, 1), "random")])
def test_two_alternating_tables_take_the_single_pass_kernel(gpu, oracle, mcu, other):
    rng = np.random.default_rng(77 + mcu[0])
    W, H = 4096, 768
    tab_b = C.ALT if other == "alt" else C.random_huffman_table(rng, n_cat=16, skew=1.5)
    n = mcu[0]
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=mcu, tables=(C.NIKON, tab_b),
        table_index=[0, 1] * (n // 2))
    assert d.n_tables == 2
    plan, inp, out = _run(gpu, oracle, d, data, W, H)
    import bench_ljpeg as B
    assert np.array_equal(B.gpu_frame(out, 0, W, H), tile_px)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names), names
    assert not any("sync" in x for x in names), names


def test_table_order_b_a(gpu, oracle):
    """The component of the even symbols may use DHT slot 1 and the other one slot 0."""
    rng = np.random.default_rng(5)
    W, H = 2048, 512
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1), tables=(C.ALT, C.NIKON),
        table_index=[1, 0])
    plan, inp, out = _run(gpu, oracle, d, data, W, H)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names), names


def test_not_alternating_goes_through_the_pipeline(gpu, oracle):
    """A A A B over four components is not the A B A B the kernel knows: the multi-kernel
    pipeline decodes it, bit-exactly."""
    rng = np.random.default_rng(6)
    W, H = 2048, 256
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(4, 1), tables=(C.NIKON, C.ALT),
        table_index=[0, 0, 0, 1])
    plan, inp, out = _run(gpu, oracle, d, data, W, H)
    names = _kernel_names(plan, inp, out)
    assert not any("lj_fast_kernel" in x for x in names), names


@pytest.mark.parametrize("seed", range(4))
def test_long_codes_and_ssss16_in_either_table(gpu, oracle, seed):
    """Random canonical tables with codes longer than the 10-bit LUTs and full-range data
    (SSSS = 16): lanes stop at such symbols and are re-decoded the general way, with the
    table their position in the stream names."""
    rng = np.random.default_rng(900 + seed)
    W, H = 1536, 256
    ta = C.random_huffman_table(rng)
    tb = C.random_huffman_table(rng)
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1), tables=(ta, tb),
        table_index=[0, 1], prec=16, full_range=True)
    _run(gpu, oracle, d, data, W, H)


def test_nearly_identical_tables_stay_exact(gpu, oracle):
    """Two tables that differ in ONE rare symbol: a parse with the tables swapped follows
    the true one for long stretches, so the table bit of the start guesses does not
    synchronise.  Whatever route the library ends up taking, the pixels are the
    reference's."""
    rng = np.random.default_rng(8)
    W, H = 2048, 384
    counts, values = C.NIKON
    v2 = list(values)
    v2[-1], v2[-2] = v2[-2], v2[-1]
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1),
        tables=(C.NIKON, (counts, v2)), table_index=[0, 1])
    assert d.n_tables == 2
    _run(gpu, oracle, d, data, W, H)


def test_constant_regions_with_two_tables(gpu, oracle):
    """Clipped highlights: the two zero-difference codes in turn, for whole workgroups."""
    rng = np.random.default_rng(9)
    W, H = 4096, 512
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1), tables=(C.NIKON, C.ALT),
        table_index=[0, 1])
    # (re-encode with a clipped image: the case builder's image, its top third constant)
    tile_px = tile_px.copy()
    tile_px[: H // 3, :] = 16383
    tile_px[H // 2: H // 2 + 40, 1000:3000] = 0
    rows = C.ljpeg_stream_rows(tile_px, 2, 1, W // 2, H, rng, 14)
    scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.ALT])
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    _run(gpu, oracle, d, data, W, H)


def test_many_workgroups_and_tiles_with_two_tables(gpu, oracle):
    """Four DNG-style tiles of 2048 x 1024 in one plan (hundreds of workgroups per stream,
    several streams in flight), every tile with its own pair of tables."""
    import bench_ljpeg as B
    from oracle_lib import HostImage
    rng = np.random.default_rng(10)
    TW, TH = 2048, 1024
    W, H = 2 * TW, 2 * TH
    op = B.out_pitch(W)
    want = HostImage(W, H)
    jobs, blobs, off = [], [], 0
    for k, (tx, ty) in enumerate([(0, 0), (TW, 0), (0, TH), (TW, TH)]):
        tb = C.ALT if k % 2 == 0 else C.random_huffman_table(rng, n_cat=16, skew=1.2)
        d, data, _, _ = C.make_ljpeg_case(
            rng, img_w=W, img_h=H, cpp=1, tile=(tx, ty, TW, TH), mcu=(2, 1),
            tables=(C.NIKON, tb), table_index=[0, 1])
        st_o, _ = oracle.ljpeg(d, data, want)
        assert st_o == 0
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
        jobs.append(j)
        pad = (-data.size) % 16
        blobs.append(np.concatenate([data, np.zeros(pad, np.uint8)]))
        off += data.size + pad
    plan = gpu.ljpeg_plan(jobs)
    inp = torch.from_numpy(np.concatenate(blobs)).cuda()
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, cons = plan.results()
    assert rc == 0 and not any(st)
    px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
    assert np.array_equal(px, want.pixels())
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names), names
    assert not any("sync" in x for x in names), names
