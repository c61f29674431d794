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


def test_not_alternating_takes_the_table_per_phase_instantiation(gpu, oracle):
    """A A A B over four components is not the A B A B of the two-table instantiation: until
    round 5 the multi-kernel pipeline decoded it; since round 6 it is a table-per-phase stream
    (tests/test_gpu_per_component_tables.py) of the single-pass kernel."""
    rng = np.random.default_rng(6)
    W, H = 2048, 256
    d, data, tile_px, _ = C.make_ljpeg_case(
        rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(4, 1), tables=(C.NIKON, C.ALT),
        table_index=[0, 0, 0, 1])
    plan, inp, out = _run(gpu, oracle, d, data, W, H)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names), names
    assert not any("sync" in x for x in names), names


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


# a second code whose longest word has 10 bits (15 categories, 14-bit data)
SHORT = ([0, 2, 2, 2, 2, 1, 1, 1, 1, 3, 0, 0, 0, 0, 0, 0],
         [6, 7, 5, 8, 4, 9, 3, 10, 2, 11, 1, 12, 0, 13, 14])


@pytest.mark.parametrize("n_comp,slices", [(2, (3, 1344, 1408)), (4, (2, 2048, 2048))])
def test_cr2_with_two_tables(gpu, oracle, n_comp, slices):
    """Cr2Decompressor <N,1,1> whose components take DHT slots 0, 1 (, 0, 1): the strips'
    copy-out behind the two-table decode.  (Codes of at most 10 bits: where a strip row
    ends the stream jumps across the image, the differences there are large, and a table
    that gives the large categories codes the 10-bit LUTs do not hold stops a lane at every
    such jump -- more of them than a workgroup has re-decode entries, and the stream goes
    through the multi-kernel pipeline: the next test.)"""
    from oracle_lib import HostImage
    rng = np.random.default_rng(31 + n_comp)
    W = (slices[0] - 1) * slices[1] + slices[2]
    H = 900
    d, data, img, scan_len = C.make_cr2_case(rng, W, H, n_comp, slices, tables=(C.NIKON, SHORT),
                                             table_index=[0, 1] * (n_comp // 2))
    want, got = HostImage(W, H), HostImage(W, H)
    st_o, cons_o = oracle.cr2(d, data, want)
    assert st_o == 0 and np.array_equal(want.pixels(), img)
    st, cons = gpu.cr2_decode(d, data, got.view())
    assert (st, cons) == (st_o, cons_o)
    assert np.array_equal(got.u16(), want.u16())
    # the route, through a plan
    j = abi.Cr2Job()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    import bench_ljpeg as B
    op = B.out_pitch(W)
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
    plan = gpu.cr2_plan([j])
    inp = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st2, cons2 = plan.results()
    assert rc == 0 and not any(st2) and list(cons2) == [cons_o]
    px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
    assert np.array_equal(px, img)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names), names
    assert not any("sync" in x for x in names), names


def test_cr2_two_tables_long_codes_at_every_strip_row(gpu, oracle):
    """The same with a table whose large categories have codes of 11-16 bits: whatever the
    route, the reference's pixels."""
    from oracle_lib import HostImage
    rng = np.random.default_rng(33)
    slices = (3, 1344, 1408)
    W, H = 2 * 1344 + 1408, 600
    d, data, img, scan_len = C.make_cr2_case(rng, W, H, 2, slices, tables=(C.NIKON, C.ALT),
                                             table_index=[0, 1])
    want, got = HostImage(W, H), HostImage(W, H)
    st_o, cons_o = oracle.cr2(d, data, want)
    assert st_o == 0
    assert gpu.cr2_decode(d, data, got.view()) == (st_o, cons_o)
    assert np.array_equal(got.u16(), want.u16())


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_two_tables(gpu, oracle, seed):
    """Differential fuzzing of the two-table instantiation: images stitched from noise,
    constant and clipped stretches, ramps and short periods (test_gpu_fast_fuzz's), two random
    canonical tables (long codes, SSSS = 16 with 16-bit data), 2 or 4 components, tiles
    narrower than their frames, bytes behind the end-of-image marker, several streams a call.
    Whatever route a stream takes, status, consumed bytes and pixels are the oracle's."""
    from oracle_lib import HostImage
    from test_gpu_fast_fuzz import banded_image
    rng = np.random.default_rng([3031, seed])
    n = int(rng.choice([2, 2, 4]))
    prec = int(rng.choice([12, 14, 14, 16]))
    n_cat = 17 if prec == 16 else prec + 1
    ta = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
    tb = C.random_huffman_table(rng, n_cat, skew=float(rng.uniform(0.4, 2.5)))
    k = int(rng.integers(1, 4))
    tiles, x = [], 0
    H = int(rng.integers(120, 500))
    for _ in range(k):
        tw = n * int(rng.integers(40, 1400 // n))
        tiles.append((x, tw))
        x += tw
    W = x + int(rng.integers(0, 9))
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas, pxs = [], [], []
    for tx, tw in tiles:
        th = H - int(rng.integers(0, 3))
        px = banded_image(rng, th, tw, prec)
        fw = (tw + n - 1) // n + int(rng.integers(0, 3))
        rows = C.ljpeg_stream_rows(px, n, 1, fw, th, rng, prec)
        init = [1 << (prec - 1)] * n
        order = [0, 1] if rng.integers(0, 2) else [1, 0]
        idx = order * (n // 2)
        scan, _ = synth.ljpeg_encode_scan(rows, n, init, [(ta, tb)[i] for i in idx], 0, False)
        d = abi.LJpegDesc()
        d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, 0, tw, th
        d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = n, 1, fw, th
        d.n_comp, d.rows_per_restart_interval = n, th
        abi.fill_recipe(d, synth.huff_tables(ta, tb), idx, init)
        tail = int(rng.integers(0, 3))
        extra = {0: np.zeros(16, np.uint8), 1: np.zeros(int(rng.integers(16, 40000)), np.uint8),
                 2: rng.integers(0, 256, int(rng.integers(16, 40000)), dtype=np.uint8)}[tail]
        descs.append(d)
        datas.append(np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), extra]))
        pxs.append(px)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    for i in range(k):
        assert st[i] == so[i][0], (i, st, so)
        if so[i][0] == 0:
            assert cons[i] == so[i][1], (i, cons, so)
    if all(s[0] == 0 for s in so):
        assert np.array_equal(img.u16(), want.u16())
        for (tx, tw), px in zip(tiles, pxs):
            assert np.array_equal(img.pixels()[:px.shape[0], tx:tx + tw], px)


def _tile_job(d, data, off, W, H, op):
    j = abi.LJpegJob()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
    return j


def test_mixed_plan_one_table_two_tables_and_a_pipeline_stream(gpu, oracle):
    """One plan, three kinds of streams side by side: a one-table tile, a two-table tile (K0's
    two-table instantiation and its hand-over then serve the one-table stream as well) and a
    tile with a 2 x 2 MCU, which the multi-kernel pipeline decodes.  (Until round 5 the third was
    a 3-component tile with tables A B A: a table-per-phase stream of the single-pass kernel now.)"""
    import bench_ljpeg as B
    from oracle_lib import HostImage
    rng = np.random.default_rng(12)
    TW, TH = 1536, 640
    W, H = 3 * TW, TH
    op = B.out_pitch(W)
    want = HostImage(W, H)
    cases = [
        dict(mcu=(2, 1), tables=(C.NIKON,), table_index=[0, 0]),
        dict(mcu=(2, 1), tables=(C.NIKON, C.ALT), table_index=[0, 1]),
        dict(mcu=(2, 2), tables=(C.NIKON, C.ALT), table_index=[0, 1, 0, 1]),
    ]
    jobs, blobs, off = [], [], 0
    for k, kw in enumerate(cases):
        d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(k * TW, 0, TW, TH), **kw)
        assert oracle.ljpeg(d, data, want)[0] == 0
        jobs.append(_tile_job(d, data, off, W, H, op))
        pad = (-data.size) % 16
        blobs.append(np.concatenate([data, np.zeros(pad, np.uint8)]))
        off += data.size + pad
    plan = gpu.ljpeg_plan(jobs)
    inp = torch.from_numpy(np.concatenate(blobs)).cuda()
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rc, st, cons = plan.results()
        assert rc == 0 and not any(st)
        px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
        assert np.array_equal(px, want.pixels())
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in x for x in names) and any("sync" in x for x in names), names


def test_plan_reused_with_other_data(gpu, oracle):
    """The same plan over different inputs in turn (A, B, A, B, B): what K0's workgroups hand
    one another -- entry states tagged with the run's parity -- and what the single-pass
    kernel's workgroups leave for their successors is all per run."""
    import bench_ljpeg as B
    from oracle_lib import HostImage
    W, H = 3072, 1024
    op = B.out_pitch(W)
    made = []
    for seed in (21, 22):
        rng = np.random.default_rng(seed)
        tile_px = C.smooth_image(rng, H, W, 14, sigma=30.0 if seed == 21 else 6.0)
        rows = C.ljpeg_stream_rows(tile_px, 2, 1, W // 2, H, rng, 14)
        scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.ALT])
        made.append((tile_px, scan))
    # (one descriptor for both: the inputs are padded to the same length)
    n = max(len(s) for _, s in made) + 2 + 4096
    datas = []
    for _, scan in made:
        data = np.zeros(n, np.uint8)
        data[:len(scan)] = scan
        data[len(scan):len(scan) + 2] = (0xFF, 0xD9)
        datas.append(data)
    rng = np.random.default_rng(5)
    d, _, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1),
                                   tables=(C.NIKON, C.ALT), table_index=[0, 1])
    wants = []
    for data in datas:
        w = HostImage(W, H)
        assert oracle.ljpeg(d, data, w)[0] == 0
        wants.append(w.pixels().copy())
    assert np.array_equal(wants[0], made[0][0]) and np.array_equal(wants[1], made[1][0])
    plan = gpu.ljpeg_plan([_tile_job(d, datas[0], 0, W, H, op)])
    ins = [torch.from_numpy(x).cuda() for x in datas]
    out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
    for k in (0, 1, 0, 1, 1):
        out.zero_()
        plan.run(ins[k].data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rc, st, cons = plan.results()
        assert rc == 0 and not any(st)
        px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
        assert np.array_equal(px, wants[k]), k


def test_host_pointer_dng_call_in_bands_of_tile_rows(gpu, oracle, monkeypatch):
    """A large rsx_dng_decompress_ljpeg call (host pointers, >= 8 MB of tiles) runs as up to
    four bands of tile rows on helper threads, taking turns on each direction of the link
    (rsx_api.hip).  Same statuses, consumed bytes and pixels as the call in one piece -- with
    one damaged tile in the third row, whose status must come back in ITS slot -- and the
    context counts ONE host call."""
    from oracle_lib import HostImage
    from rawspeed_amd import capi
    rng = np.random.default_rng(77)
    TW, TH, NX, NY = 2048, 1024, 2, 4
    W, H = TW * NX, TH * NY
    descs, datas = [], []
    want = HostImage(W, H)
    for ty in range(NY):
        for tx in range(NX):
            two = (tx + ty) % 2 == 0
            d, data, _, scan_len = C.make_ljpeg_case(
                rng, img_w=W, img_h=H, cpp=1, tile=(tx * TW, ty * TH, TW, TH), mcu=(2, 1),
                tables=(C.NIKON, C.ALT) if two else (C.NIKON,),
                table_index=[0, 1] if two else [0, 0], sigma=60.0)
            if (tx, ty) == (1, 2):       # cut short: the symbols run out of data
                data = np.concatenate([data[:scan_len // 3], np.zeros(64, np.uint8)])
            descs.append(d)
            datas.append(data)
    assert sum(x.size for x in datas) >= 8 << 20
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    assert so[5][0] != 0 and all(s[0] == 0 for i, s in enumerate(so) if i != 5)
    calls0 = gpu.host_calls() if hasattr(gpu, "host_calls") else None
    img = HostImage(W, H)
    rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    assert rc == abi.RSX_ERR_TILE_ERRORS
    assert list(st) == [s[0] for s in so]
    assert all(c == s[1] for c, s in zip(cons, so) if s[0] == 0)
    if calls0 is not None:
        assert gpu.host_calls() == calls0 + 1
    # every good tile's rectangle is the oracle's
    for i, d in enumerate(descs):
        if so[i][0] == 0:
            y, x = d.tile_y, d.tile_x
            assert np.array_equal(img.pixels()[y:y + TH, x:x + TW], want.pixels()[y:y + TH, x:x + TW])
    monkeypatch.setenv("RSX_HOST_NO_OVERLAP", "1")
    one_piece = capi.Context(0)
    ref = HostImage(W, H)
    rc2, st2, cons2 = one_piece.dng_decompress_ljpeg(descs, datas, ref.view())
    assert (rc2, list(st2)) == (rc, list(st))
    for i, d in enumerate(descs):
        if so[i][0] == 0:
            y, x = d.tile_y, d.tile_x
            assert np.array_equal(img.pixels()[y:y + TH, x:x + TW], ref.pixels()[y:y + TH, x:x + TW])
