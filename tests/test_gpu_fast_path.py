"""GPU parity of the single-pass LJPEG kernel (rsx_ljpeg_fast.hip) on the shapes that
exercise ITS machinery -- the look-backs, the row table, the staging capacity, the
re-decode rounds, the hand-over to the multi-kernel pipeline -- through the C-ABI against
the oracle.  (The general LJPEG / CR2 / DNG parity tests run through it as well: every
single-table stream with 1, 2 or 4 components in a row of an MCU takes it.)"""
import numpy as np
import pytest
import torch

from rawspeed_amd import abi, synth

import cases as C
from oracle_lib import HostImage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import gpu_util
    return gpu_util.ctx()


def _check(gpu, oracle, d, data, w, h, cpp=1):
    img, want = HostImage(w, h, cpp), HostImage(w, h, cpp)
    so = oracle.ljpeg(d, data, want)
    sg = gpu.ljpeg_decode(d, data, img.view())
    assert sg == so, (sg, so)
    if so[0] == 0:
        assert np.array_equal(img.u16(), want.u16())
    return so


@pytest.mark.parametrize("tw,n", [(256, 2), (128, 2), (64, 2), (32, 2), (16, 1), (48, 4), (8, 2)])
def test_narrow_tiles_many_rows_per_workgroup(gpu, oracle, tw, n):
    """A workgroup (15 000 symbols) of a narrow tile holds up to 256 stream rows in the
    kernel's row table; narrower than that, the stream goes to the multi-kernel pipeline.
    Either way the pixels are the reference's."""
    rng = np.random.default_rng([7, tw, n])
    H = 1200 if tw >= 64 else 2400
    d, data, px, _ = C.make_ljpeg_case(rng, img_w=tw + 10, img_h=H, cpp=1, tile=(3, 0, tw, H),
                                       mcu=(n, 1))
    so = _check(gpu, oracle, d, data, tw + 10, H)
    assert so[0] == 0


@pytest.mark.parametrize("sigma", [0.3, 1.0, 3.0])
def test_low_entropy_streams(gpu, oracle, sigma):
    """Few bits per symbol: more than 128 symbols in a 64-byte subsequence (the registers
    a lane keeps) and more samples per workgroup than its LDS stages -- the plan sizes the
    allocation from the stream, or keeps the stream off the single-pass kernel, or a
    workgroup hands it over."""
    rng = np.random.default_rng([8, int(sigma * 10)])
    W, H = 2048, 600
    d, data, px, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H),
                                       mcu=(2, 1), sigma=sigma)
    so = _check(gpu, oracle, d, data, W, H)
    assert so[0] == 0


def test_flat_then_noisy_image_hands_over(gpu, oracle):
    """An image whose first rows are noisy and whose rest is nearly flat: the allocation
    sized from the stream's average does not hold the flat part's workgroups."""
    rng = np.random.default_rng(9)
    W, H = 2048, 800
    px = C.smooth_image(rng, H, W, sigma=30.0)
    flat = C.smooth_image(rng, H, W, sigma=0.4)
    px[H // 4:] = flat[H // 4:]
    rows = C.ljpeg_stream_rows(px, 2, 1, W // 2, H, rng)
    scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [C.NIKON, C.NIKON], 0, False)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = 0, 0, W, H
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = 2, 1, W // 2, H
    d.n_comp, d.rows_per_restart_interval = 2, H
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0, 0], [1 << 13] * 2)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    so = _check(gpu, oracle, d, data, W, H)
    assert so[0] == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_tables_with_long_codes(gpu, oracle, seed):
    """Random canonical tables: codes longer than the 10-bit LUT and SSSS = 16 stop a lane
    of the fast loop; those subsequences are re-decoded with the general step."""
    rng = np.random.default_rng([10, seed])
    counts, values = C.random_huffman_table(rng, n_cat=17, skew=float(rng.uniform(0.5, 2.5)))
    W, H = 1024, 300
    d, data, px, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H),
                                       mcu=(2, 1), tables=((counts, values),), prec=16,
                                       full_range=bool(seed & 1), fix16=bool(seed & 2))
    _check(gpu, oracle, d, data, W, H)


def test_many_streams_of_unequal_length_interleaved(gpu, oracle):
    """One batched call, tiles of very different sizes: the workgroups of all streams
    take their tickets interleaved, each stream's in order."""
    rng = np.random.default_rng(11)
    W, H = 3000, 700
    img, want = HostImage(W, H), HostImage(W, H)
    descs, datas = [], []
    x = 0
    for k, tw in enumerate((2048, 16, 512, 128, 256, 32)):
        th = (700, 80, 300, 700, 33, 500)[k]
        d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(x, 0, tw, th),
                                          mcu=(2, 1))
        descs.append(d)
        datas.append(data)
        x += tw
    cons = []
    for d, data in zip(descs, datas):
        st, c = oracle.ljpeg(d, data, want)
        assert st == 0
        cons.append(c)
    rc, st, got = gpu.dng_decompress_ljpeg(descs, datas, img.view())
    assert rc == 0 and not any(st) and got == cons
    assert np.array_equal(img.u16(), want.u16())


def test_damaged_streams_in_a_batch_leave_the_others_alone(gpu, oracle):
    """A truncated stream and one with an invalid code among healthy ones, every run of the
    same plan: the damaged streams are handed to the multi-kernel pipeline (and demoted
    after two runs), the healthy ones keep their pixels."""
    rng = np.random.default_rng(12)
    W, H, tw = 2048, 400, 512
    descs, datas = [], []
    for k in range(4):
        d, data, _, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(k * tw, 0, tw, H),
                                          mcu=(2, 1))
        descs.append(d)
        datas.append(data)
    datas[1] = datas[1][:len(datas[1]) // 2]
    bad = datas[3].copy()
    bad[len(bad) // 3:len(bad) // 3 + 8] = 0xFF
    bad[len(bad) // 3 + 1:len(bad) // 3 + 8:2] = 0xFE
    datas[3] = bad
    want = HostImage(W, H)
    so = [oracle.ljpeg(d, data, want) for d, data in zip(descs, datas)]
    assert so[0][0] == 0 and so[2][0] == 0
    for run in range(4):
        img = HostImage(W, H)
        rc, st, cons = gpu.dng_decompress_ljpeg(descs, datas, img.view())
        for k in range(4):
            assert (st[k] != 0) == (so[k][0] != 0), (run, k, st, so)
            if so[k][0] == 0:
                assert cons[k] == so[k][1]
                assert np.array_equal(img.pixels()[:, k * tw:(k + 1) * tw],
                                      want.pixels()[:, k * tw:(k + 1) * tw]), (run, k)


def test_plan_reruns_are_identical(gpu):
    """The look-back records and tickets start from zero in every run of a plan."""
    import gpu_util
    import bench_ljpeg as B
    W, H = 1536, 512
    made = [B.make_cr2_frame(W, H, (3, 512, 512), seed=50 + f) for f in range(3)]
    plan, inp, out = B._cr2_batch(gpu, torch, [(m[0], m[1]) for m in made], W, H)
    s = torch.cuda.current_stream().cuda_stream
    ref = None
    for run in range(5):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and list(cons) == [m[3] for m in made]
        got = out.cpu().numpy().copy()
        for f in range(3):
            assert np.array_equal(B.gpu_frame(out, f, W, H), made[f][2])
        if ref is None:
            ref = got
        assert np.array_equal(got, ref)


def _kernel_names(plan, inp, out):
    s = torch.cuda.current_stream().cuda_stream
    plan.set_timing(True)
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    tab = plan.kernel_table()
    plan.set_timing(False)
    return [n for n, _ in tab[0]] if tab else []


@pytest.mark.parametrize("W,H", [(2048, 768), (4480, 1024)])
def test_constant_regions_stay_on_the_single_pass_kernel(gpu, W, H):
    """Blown highlights and a black border: inside them the stream is the code of the zero
    difference over and over -- no parse from an arbitrary bit synchronises there, and a
    workgroup holds six times the symbols of sensor noise.  K0 reads the symbol grid of
    such slots from their bits and picks the LDS level of the run; the multi-kernel
    pipeline (its synchronisation kernel) is not launched at all."""
    import bench_ljpeg as B
    from rawspeed_amd import abi, synth
    made = []
    for f in range(2):
        src = B.clipped_image(W, H, 70 + f)
        rows = C.cr2_stream_from_image(src, 2, W // 2, H, C.cr2_slices(2, W // 2, W // 2))
        scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()])
        d = abi.Cr2Desc()
        d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
        d.frame_w, d.frame_h = W // 2, H
        d.num_slices, d.slice_width, d.last_slice_width = 2, W // 2, W // 2
        abi.fill_recipe(d, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
        pad = (-(len(scan) + 2)) % 16 + 16
        data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
        made.append((d, data, src, len(scan)))
    plan, inp, out = B._cr2_batch(gpu, torch, [(m[0], m[1]) for m in made], W, H)
    s = torch.cuda.current_stream().cuda_stream
    for run in range(3):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and list(cons) == [m[3] for m in made]
        for f in range(2):
            assert np.array_equal(B.gpu_frame(out, f, W, H), made[f][2]), (run, f)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in n for n in names), names
    assert not any("sync" in n for n in names), names


def test_lds_level_follows_the_data_of_a_reused_plan(gpu):
    """One plan, first a noisy frame, then (same byte count) a frame with constant regions:
    the host has seen that the plan's data needs the smallest LDS level only and launches
    no other, so the run with the dense frame hands its streams to the multi-kernel pipeline
    -- same pixels -- and the run after that launches the level the data asked for."""
    import bench_ljpeg as B
    from rawspeed_amd import abi, synth
    W, H = 4480, 1024

    def encode(src):
        rows = C.cr2_stream_from_image(src, 2, W // 2, H, C.cr2_slices(2, W // 2, W // 2))
        scan, _ = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [B._nikon(), B._nikon()])
        return scan

    noisy = synth.sensor_image(W, H, 14, seed=90)
    dense = B.clipped_image(W, H, 91)
    sa, sb = encode(noisy), encode(dense)
    assert len(sb) < len(sa)
    n = len(sa) + 2 + ((-(len(sa) + 2)) % 16 + 16)

    def blob(scan):
        out = np.zeros(n, np.uint8)
        out[:len(scan)] = scan
        out[len(scan):len(scan) + 2] = (0xFF, 0xD9)
        return out

    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = 2, W // 2, W // 2
    abi.fill_recipe(d, synth.huff_tables(B._nikon()), [0, 0], [1 << 13] * 2)
    plan, inp, out = B._cr2_batch(gpu, torch, [(d, blob(sa))], W, H)
    s = torch.cuda.current_stream().cuda_stream

    def run(scan, src):
        inp.copy_(torch.from_numpy(blob(scan)))
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and list(cons) == [len(scan)]
        assert np.array_equal(B.gpu_frame(out, 0, W, H), src)

    run(sa, noisy)
    run(sa, noisy)
    run(sb, dense)
    run(sb, dense)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel(" in n for n in names), names
    assert not any("sync" in n for n in names), names
    run(sa, noisy)


@pytest.mark.parametrize("tail", ["zeros", "pattern", "random"])
def test_trailing_bytes_after_the_last_symbol_are_nobodys_business(gpu, oracle, tail):
    """200 KB behind the end-of-image marker inside the job's byte range (the bit pump
    feeds zeros there; the kernel's slots are periodic, nothing synchronises): no delivered
    symbol lies in them, the stream stays on the single-pass kernel."""
    import bench_ljpeg as B
    from rawspeed_amd import abi
    rng = np.random.default_rng([13, len(tail)])
    W, H = 2048, 700
    d, data, px, _ = C.make_ljpeg_case(rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1))
    extra = {"zeros": np.zeros(200000, np.uint8),
             "pattern": np.tile(np.array([0x55, 0xAA, 0x3C], np.uint8), 70000),
             "random": rng.integers(0, 256, 200000, dtype=np.uint8)}[tail]
    data = np.concatenate([data, extra])
    so = _check(gpu, oracle, d, data, W, H)
    assert so[0] == 0
    j = abi.LJpegJob()
    j.desc = d
    j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
        B.out_pitch(W), W, H, 1, 1
    plan = gpu.ljpeg_plan([j])
    inp = torch.from_numpy(data).cuda()
    out = torch.zeros(B.out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for run in range(3):
        out.zero_()
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and not any(st) and list(cons) == [so[1]]
        assert np.array_equal(B.gpu_frame(out, 0, W, H), px)
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in n for n in names), names
    assert not any("sync" in n for n in names), names


def test_identical_tables_in_two_dht_slots_take_the_single_pass_kernel(gpu, oracle):
    """The reference binds a decoder per DHT slot (AbstractLJpegDecoder.h:112-125); writers
    commonly declare the same code twice, once per component.  The library compares table
    CONTENTS: such a stream is a one-table stream for the kernels (single-pass kernel, no
    synchronisation kernel); two different codes take the kernel's two-table instantiation
    (tests/test_gpu_two_tables.py)."""
    import bench_ljpeg as B
    from oracle_lib import HostImage
    rng = np.random.default_rng(4242)
    W, H = 2048, 512
    other = C.random_huffman_table(rng, n_cat=16)
    for tables, expect_fast in (((C.NIKON, C.NIKON), True), ((C.NIKON, other), True)):
        d, data, tile_px, scan_len = C.make_ljpeg_case(
            rng, img_w=W, img_h=H, cpp=1, tile=(0, 0, W, H), mcu=(2, 1), tables=tables,
            table_index=[0, 1])
        assert d.n_tables == 2
        want = HostImage(W, H)
        st_o, cons_o = oracle.ljpeg(d, data, want)
        assert st_o == 0
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = 0, data.size, 0
        op = B.out_pitch(W)
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = op, W, H, 1, 1
        plan = gpu.ljpeg_plan([j])
        inp = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        out = torch.zeros(op * H, dtype=torch.uint8, device="cuda")
        plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rc, st, cons = plan.results()
        assert rc == 0 and list(cons) == [cons_o]
        px = out.cpu().numpy().view(np.uint16).reshape(H, op // 2)[:, :W]
        assert np.array_equal(px, want.pixels()) and np.array_equal(px, tile_px)
        names = _kernel_names(plan, inp, out)
        assert any("lj_fast_kernel" in n for n in names) == expect_fast, names
        assert any("sync" in n for n in names) == (not expect_fast), names


def _cr2_general_case(rng, W, H, n_comp, n_slices, slice_w, last_w, frame_h):
    """<N,1,1> CR2 stream of a W x H (samples x rows) image whose LJPEG frame is frame_h rows
    tall -- frame_h != H: the slices wrap to the next output column
    (Cr2DecompressorImpl.h:104-154) -- cut into n_slices slices (widths in samples)."""
    widths = [slice_w] * (n_slices - 1) + [last_w]
    img = C.smooth_image(rng, H, W, 14)
    tiles = C.cr2_output_tiles([w // n_comp for w in widths], W // n_comp, H, frame_h)
    flat = np.concatenate([img[y:y + h, x * n_comp:(x + w) * n_comp].reshape(-1)
                           for x, y, w, h in tiles])
    assert flat.size == W * H, (flat.size, W * H)
    frame_w = W * H // (frame_h * n_comp)
    assert frame_w * frame_h * n_comp == W * H
    rows = np.ascontiguousarray(flat.reshape(frame_h, frame_w * n_comp))
    init_pred = [1 << 13] * n_comp
    scan, _ = synth.ljpeg_encode_scan(rows, n_comp, init_pred, [C.NIKON] * n_comp)
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = n_comp, 1, 1
    d.frame_w, d.frame_h = frame_w, frame_h
    d.num_slices, d.slice_width, d.last_slice_width = n_slices, slice_w, last_w
    abi.fill_recipe(d, synth.huff_tables(C.NIKON), [0] * n_comp, init_pred)
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(16, np.uint8)])
    return d, data, img, len(scan)


@pytest.mark.parametrize("shape", [
    # W, H, N, slices, slice_w, last_w, frame_h
    (3000, 1200, 2, 4, 1000, 1000, 900),    # wrapped slices: frame.y < dim.y
    (3000, 1200, 2, 3, 1500, 1500, 800),    # three wide slices over two columns
    (6000, 700, 2, 3, 2208, 1584, 700),     # unequal last slice
    (4096, 900, 4, 4, 1024, 1024, 900),     # <4,1,1>
    (4096, 900, 4, 4, 2048, 2048, 450),     # <4,1,1>, wrapped
], ids=["wrapped", "wrapped_wide", "unequal_last", "n4", "n4_wrapped"])
def test_cr2_strip_copy_out_at_hundreds_of_workgroups(gpu, oracle, shape):
    """The single-pass kernel's CR2 copy-out (runs cut at strip rows AND stream rows, strips
    that change inside a workgroup, slices that wrap to the next output column) at sizes
    where a stream is hundreds of workgroups, against the oracle and -- where it is built --
    the unmodified reference (Cr2DecompressorImpl.h:121-205)."""
    import bench_ljpeg as B
    from oracle_lib import Ref
    W, H, N, ns, sw, lw, fh = shape
    rng = np.random.default_rng([606, W, H, N, fh])
    d, data, img, scan_len = _cr2_general_case(rng, W, H, N, ns, sw, lw, fh)
    want = HostImage(W, H)
    st_o, cons_o = oracle.cr2(d, data, want)
    assert st_o == 0 and cons_o == scan_len
    assert np.array_equal(want.pixels(), img)
    if Ref.available():
        r = Ref()
        rimg = r.image(W, H, 1)
        st_r, cons_r = r.cr2(d, data, rimg)
        assert st_r == 0 and cons_r == scan_len and np.array_equal(rimg.pixels(), img)
    plan, inp, out = B._cr2_batch(gpu, torch, [(d, data), (d, data)], W, H)
    plan.run(inp.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    rc, st, cons = plan.results()
    assert rc == 0 and list(cons) == [scan_len, scan_len]
    for f in range(2):
        assert np.array_equal(B.gpu_frame(out, f, W, H), img), f
    assert data.size > 100 * 255 * 64   # hundreds of workgroups a stream
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in n for n in names), names
    assert not any("sync" in n for n in names), names


# ---------------------------------------------------------------------------------------
# Three interleaved components (MCU 3 x 1, linear DNG: LJpegDecompressor.cpp:102-105) on the
# single-pass kernel's <3> instantiation (round 5)
# ---------------------------------------------------------------------------------------
def _three_comp_plan(gpu, shapes, seed, **kw):
    """one job per (img_w, img_h, tile) with cpp = 3, MCU 3 x 1; returns plan, buffers, cases"""
    from oracle_lib import HostImage
    rng = np.random.default_rng(seed)
    jobs, parts, made, off, ooff = [], [], [], 0, 0
    for (w, h, tile, frame) in shapes:
        d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=3, tile=tile,
                                                       mcu=(3, 1), frame=frame, **kw)
        pad = (-data.size) % 16
        data = np.concatenate([data, np.zeros(pad, np.uint8)])
        img = HostImage(w, h, 3, is_cfa=False)
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes = off, data.size - pad
        j.img_offset = ooff
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            img.pitch, w, h, 3, 0
        jobs.append(j)
        parts.append(data)
        made.append((d, data[:data.size - pad], img, scan_len))
        off += data.size
        ooff += img.pitch * h
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.full((ooff,), 0xA5, dtype=torch.uint8, device="cuda")
    return gpu.ljpeg_plan(jobs), inp, out, made


@pytest.mark.parametrize("shapes", [
    [(96, 40, (0, 0, 96, 40), None)],                       # one small tile
    [(1500, 700, (0, 0, 1500, 700), None)],                 # hundreds of workgroups
    [(640, 300, (64, 20, 500, 250), (520, 250))],           # a tile inside the image, trailing MCUs of the frame dropped
    [(2048, 512, (0, 0, 2048, 512), None), (333, 77, (0, 0, 333, 77), None),
     (1024, 1024, (0, 0, 1024, 1024), None)],               # several streams, odd sizes
])
def test_three_components_on_the_single_pass_kernel(gpu, oracle, shapes):
    plan, inp, out, made = _three_comp_plan(gpu, shapes, seed=len(shapes) * 31 + shapes[0][0])
    s = torch.cuda.current_stream().cuda_stream
    for run in range(2):
        out.fill_(0xA5)
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and all(x == 0 for x in st), (rc, list(st))
        o = 0
        got = out.cpu().numpy()
        for (d, data, img, scan_len), c in zip(made, cons):
            want_st, want_c = oracle.ljpeg(d, data, img)
            assert want_st == 0 and c == want_c
            n = img.pitch * img.dim_y
            assert np.array_equal(got[o:o + n], img.buf), (run, img.dim_x, img.dim_y)
            o += n
    names = _kernel_names(plan, inp, out)
    assert any("lj_fast_kernel" in n for n in names), names
    # off the int16-difference route: no legacy decode, no reconstruction, no synchronisation
    assert not any(("decode" in n) or ("legacy" in n) or ("sync" in n) for n in names), names


def test_three_components_damaged_stream_takes_the_legacy_route(gpu, oracle):
    """A 3-component stream the single-pass kernel cannot finish (it ends early: symbols past
    the end of the data) is redone by the legacy route -- same status, consumed bytes and
    pixels as the oracle, and the healthy stream next to it is untouched by that."""
    plan, inp, out, made = _three_comp_plan(
        gpu, [(900, 400, (0, 0, 900, 400), None), (600, 200, (0, 0, 600, 200), None)], seed=77)
    # cut the first stream short: zero out its tail (the scan ends with garbage / zeros)
    d0, data0, img0, scan0 = made[0]
    cut = scan0 // 2
    host = inp.cpu().numpy().copy()
    host[cut:scan0 + 2] = 0
    data0 = data0.copy()
    data0[cut:scan0 + 2] = 0
    inp = torch.from_numpy(host).cuda()
    s = torch.cuda.current_stream().cuda_stream
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    rc, st, cons = plan.results()
    want0 = oracle.ljpeg(d0, data0, img0)
    want1 = oracle.ljpeg(made[1][0], made[1][1], made[1][2])
    assert (st[0], cons[0]) == want0 or st[0] == want0[0] != 0, ((st[0], cons[0]), want0)
    assert (st[1], cons[1]) == want1 and want1[0] == 0
    got = out.cpu().numpy()
    n0 = img0.pitch * img0.dim_y
    if want0[0] == 0:
        assert np.array_equal(got[:n0], img0.buf)
    n1 = made[1][2].pitch * made[1][2].dim_y
    assert np.array_equal(got[n0:n0 + n1], made[1][2].buf)


# ---------------------------------------------------------------------------------------
# Restart intervals laid out on the device (round 5): marker scan -> sort -> stream records
# -> K0 -> single-pass kernel in ONE go, no host round trip in between
# ---------------------------------------------------------------------------------------
def _dri_plan(gpu, shapes, seed, rows_per_ri):
    rng = np.random.default_rng(seed)
    jobs, parts, made, off, ooff = [], [], [], 0, 0
    for (w, h, cpp, mcu) in shapes:
        d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=cpp,
                                                       tile=(0, 0, w, h), mcu=mcu,
                                                       rows_per_ri=rows_per_ri)
        pad = (-data.size) % 16
        img = HostImage(w, h, cpp, is_cfa=cpp == 1)
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes, j.img_offset = off, data.size, ooff
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            img.pitch, w, h, cpp, int(cpp == 1)
        jobs.append(j)
        parts.append(np.concatenate([data, np.zeros(pad, np.uint8)]))
        made.append((d, data, img, scan_len))
        off += data.size + pad
        ooff += img.pitch * h
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.full((ooff,), 0xA5, dtype=torch.uint8, device="cuda")
    return gpu.ljpeg_plan(jobs), inp, out, made


@pytest.mark.parametrize("shapes,rows_per_ri", [
    ([(1024, 600, 1, (2, 1))], 7),                      # 86 intervals, one job
    ([(2048, 512, 1, (2, 1)), (640, 333, 1, (2, 1)), (512, 96, 1, (4, 1))], 16),   # three jobs
    ([(4096, 2732, 1, (2, 1))], 28),                    # a cfg-4 tile
])
def test_restart_intervals_are_laid_out_on_the_device(gpu, oracle, shapes, rows_per_ri):
    plan, inp, out, made = _dri_plan(gpu, shapes, 900 + rows_per_ri, rows_per_ri)
    s = torch.cuda.current_stream().cuda_stream
    for run in range(3):
        out.fill_(0xA5)
        plan.run(inp.data_ptr(), out.data_ptr(), s)
        rc, st, cons = plan.results()
        assert rc == 0 and all(x == 0 for x in st), (rc, list(st))
        got = out.cpu().numpy()
        o = 0
        for (d, data, img, scan_len), c in zip(made, cons):
            want = oracle.ljpeg(d, data, img)
            assert want[0] == 0 and c == want[1], (c, want)
            n = img.pitch * img.dim_y
            assert np.array_equal(got[o:o + n], img.buf), run
            o += n
    names = _kernel_names(plan, inp, out)
    # one table from the marker scan to the single-pass kernel: nothing was fetched in between
    assert any("lj_dri_layout" in n for n in names), names
    assert any("lj_fast_kernel" in n for n in names), names
    assert names.index(next(n for n in names if "lj_dri_layout" in n)) < \
        names.index(next(n for n in names if "lj_fast_kernel" in n))


def test_restart_interval_anomalies_fall_back_to_the_host_built_plan(gpu, oracle):
    """A marker with the wrong number, a marker missing, FFxx junk behind the scan: statuses,
    consumed bytes and pixels as the oracle's (LJpegDecompressor.cpp:283-298), whichever way
    the library got there."""
    rng = np.random.default_rng(4711)
    w, h = 768, 200
    d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=1,
                                                   tile=(0, 0, w, h), mcu=(2, 1), rows_per_ri=9)
    pos = [i for i in range(scan_len - 1) if data[i] == 0xFF and 0xD0 <= data[i + 1] <= 0xD7]
    assert len(pos) == (h + 8) // 9 - 1
    variants = {"clean": data.copy()}
    v = data.copy(); v[pos[3] + 1] = 0xD0 + ((data[pos[3] + 1] - 0xD0 + 1) % 8); variants["wrong_number"] = v
    v = data.copy(); v[pos[5]] = 0x00; variants["marker_missing"] = v
    v = np.concatenate([data, np.tile(np.array([0xFF, 0xE0], np.uint8), 3000)]); variants["junk_behind"] = v
    for name, dat in variants.items():
        img_o, img_g = HostImage(w, h), HostImage(w, h)
        want = oracle.ljpeg(d, dat, img_o)
        got = gpu.ljpeg_decode(d, dat, img_g.view())
        assert got[0] == want[0], (name, got, want)
        if want[0] == 0:
            assert got == want, (name, got, want)
            assert np.array_equal(img_g.buf, img_o.buf), name


def test_device_laid_out_plan_reused_over_different_inputs(gpu, oracle):
    """One plan, three different images of the same geometry (padded to one byte count): the
    restart markers sit elsewhere in each, so the layout kernel rewrites the child plan's
    stream records, block map and ticket order every run -- and the kernels behind it must
    see THIS run's words, not the ones a cache kept from the run before."""
    w, h, rpi = 1536, 640, 11
    cases_ = []
    for seed in (1, 2, 3):
        rng = np.random.default_rng(7000 + seed)
        d, data, tile_px, scan_len = C.make_ljpeg_case(rng, img_w=w, img_h=h, cpp=1,
                                                       tile=(0, 0, w, h), mcu=(2, 1),
                                                       rows_per_ri=rpi, sigma=8.0 * seed)
        cases_.append((d, data, scan_len))
    size = max(c[1].size for c in cases_) + 64
    size += (-size) % 16
    padded = [np.concatenate([c[1], np.zeros(size - c[1].size, np.uint8)]) for c in cases_]
    img = HostImage(w, h)
    j = abi.LJpegJob()
    j.desc = cases_[0][0]
    j.in_offset, j.in_bytes, j.img_offset = 0, size, 0
    j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = img.pitch, w, h, 1, 1
    plan = gpu.ljpeg_plan([j])
    out = torch.zeros(img.pitch * h, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        for k in (0, 1, 2, 1, 0):
            inp = torch.from_numpy(padded[k]).cuda()
            out.fill_(0x5A)
            plan.run(inp.data_ptr(), out.data_ptr(), s)
            rc, st, cons = plan.results()
            want_img = HostImage(w, h, fill=0x5A)
            want = oracle.ljpeg(cases_[k][0], padded[k], want_img)
            assert rc == 0 and (st[0], cons[0]) == want, (rep, k, st[0], cons[0], want)
            assert np.array_equal(out.cpu().numpy(), want_img.buf), (rep, k)
    inp = torch.from_numpy(padded[0]).cuda()
    names = _kernel_names(plan, inp, out)
    assert any("lj_dri_layout" in n for n in names), names
