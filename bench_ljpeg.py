"""LJPEG legs of bench.py (BASELINE configs 3 and 4), measured like the headline
number: inputs/outputs resident in HBM, one plan launch per step, hipEvent time
of the dominant kernel, and the same workload through the unmodified reference
(oracle/_ref) on the host cores."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NIKON = None


def _nikon():
    from rawspeed_amd import synth
    return (synth.NIKON14_COUNTS, synth.NIKON14_VALUES)


def out_pitch(w):
    return (w * 2 + 15) // 16 * 16


def make_cr2_frame(W, H, slices, seed):
    """cfg 3: CR2-style 2-component stream of a WxH sensor-like image."""
    from rawspeed_amd import abi, synth
    import cases
    src = synth.sensor_image(W, H, 14, seed=seed)
    sl = cases.cr2_slices(*slices)
    rows = cases.cr2_stream_from_image(src, 2, W // 2, H, sl)
    scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [_nikon(), _nikon()])
    d = abi.Cr2Desc()
    d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
    d.frame_w, d.frame_h = W // 2, H
    d.num_slices, d.slice_width, d.last_slice_width = slices
    abi.fill_recipe(d, synth.huff_tables(_nikon()), [0, 0], [1 << 13] * 2)
    pad = (-(len(scan) + 2)) % 16 + 16
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
    return d, data, src, len(scan), bits


def make_tile(src_tile, tx, ty, tw, th):
    from rawspeed_amd import abi, synth
    scan, bits = synth.ljpeg_encode_scan(np.ascontiguousarray(src_tile), 2, [1 << 13] * 2,
                                         [_nikon(), _nikon()])
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, ty, tw, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = 2, 1, tw // 2, th
    d.n_comp, d.rows_per_restart_interval = 2, th
    abi.fill_recipe(d, synth.huff_tables(_nikon()), [0, 0], [1 << 13] * 2)
    pad = (-(len(scan) + 2)) % 16 + 16
    data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
    return d, data, len(scan), bits


LAST_KERNEL_TABLE = None  # per-kernel averages (ms per run) of the last _time_plan call


def _time_plan(torch, plan, inp, out, steps, warmup):
    """(seconds per step with timing off, dominant kernel (name, ms, runs), consumed).
    The per-kernel table of LJPEG-family plans comes from 3 extra runs with an event
    after every launch; the dominant kernel is the largest entry of that table."""
    global LAST_KERNEL_TABLE
    s = torch.cuda.current_stream().cuda_stream
    plan.run(inp.data_ptr(), out.data_ptr(), s)
    rc, st, cons = plan.results()
    assert rc == 0, (rc, st)
    for _ in range(warmup):
        plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    plan.set_timing(True)
    for _ in range(3):
        plan.run(inp.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    tab = plan.kernel_table()
    LAST_KERNEL_TABLE = None
    if tab:
        LAST_KERNEL_TABLE = {n: round(ms, 4) for n, ms in tab[0]}
    kt = plan.kernel_time()
    plan.set_timing(False)
    return dt, kt, cons


def _dominant(res, kt):
    if kt:
        res["dominant_kernel"] = {"name": kt[0], "avg_ms": round(kt[1], 4)}
    if LAST_KERNEL_TABLE:
        res["kernels_ms"] = dict(LAST_KERNEL_TABLE)


def run_cfg3(ctx, torch, log, frames=8, steps=10, warmup=2):
    """configs[2]: CR2-style 6720x4480, 2 components, 3 slices; `frames` frames/step."""
    from rawspeed_amd import abi
    W, H = 6720, 4480
    d, data, src, scan_len, bits = make_cr2_frame(W, H, (3, 2240, 2240), seed=1)
    jobs, off = [], 0
    for f in range(frames):
        j = abi.Cr2Job()
        j.desc = d
        j.in_offset, j.in_bytes = off, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
        off += data.size
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    out = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.cr2_plan(jobs)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    got = out[:out_pitch(W) * H].cpu().numpy().view(np.uint16).reshape(H, out_pitch(W) // 2)[:, :W]
    exact = bool(np.array_equal(got, src)) and all(c == scan_len for c in cons)
    alg = frames * (scan_len + W * H * 2)
    res = {
        "workload": "Cr2Decompressor <2,1,1> 6720x4480, 3 slices, %d frames/step" % frames,
        "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
        "ms_per_step": round(dt * 1e3, 4),
        "bit_exact": exact,
        "entropy_bits_per_px": round(scan_len * 8 / (W * H), 3),
        "algorithmic_bytes_per_step": alg,
        "achieved_gbps_whole_pipeline": round(alg / dt / 1e9, 1),
        "frac_of_hbm_peak": round(alg / dt / 1e9 / 8000.0, 4),
    }
    _dominant(res, kt)
    return res, (d, data, W, H)


def run_sraw(ctx, torch, log, frames=8, steps=10, warmup=2):
    """Canon sRaw1/mRAW <3,2,2> of a 3960x2640 px frame: 1980 groups
    x 1320 rows x 6 samples (Y Y Y Y Cb Cr), 3 slices; samples/s is the unit."""
    import cases
    from rawspeed_amd import abi
    from rawspeed_amd import synth
    src = synth.sensor_image(1980 * 6, 1320, 14, seed=4)
    d, data, src, scan_len = cases.make_cr2_sraw_case(None, 2, (3, 660, 660), 1320, img=src)
    pad = (-data.size) % 16
    data = np.concatenate([data, np.zeros(pad, np.uint8)])
    H, W = src.shape
    jobs = []
    for f in range(frames):
        j = abi.Cr2Job()
        j.desc = d
        j.in_offset, j.in_bytes = f * data.size, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 0
        jobs.append(j)
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    out = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.cr2_plan(jobs)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    got = out[-out_pitch(W) * H:].cpu().numpy().view(np.uint16).reshape(H, out_pitch(W) // 2)[:, :W]
    exact = bool(np.array_equal(got, src)) and all(c == scan_len for c in cons)
    # Cr2sRawInterpolator on the decoded frames, still in HBM (Cr2Decoder.cpp:585-625)
    ow, oh = 2 * (W // 6), 2 * H
    opitch = (ow * 3 * 2 + 15) // 16 * 16
    sjobs = []
    for f in range(frames):
        sj = abi.SrawJob()
        sj.desc = abi.SrawDesc.make(2, 2, [2100, 1024, 1650], -80)
        sj.in_offset, sj.img_offset = f * out_pitch(W) * H, f * opitch * oh
        sj.in_.pitch_bytes, sj.in_.dim_x, sj.in_.dim_y, sj.in_.cpp = out_pitch(W), W, H, 1
        sj.img.pitch_bytes, sj.img.dim_x, sj.img.dim_y, sj.img.cpp = opitch, ow, oh, 3
        sjobs.append(sj)
    rgb = torch.empty(frames * opitch * oh, dtype=torch.uint8, device="cuda")
    splan = ctx.sraw_plan(sjobs)
    sdt, skt, _ = _time_plan(torch, splan, out, rgb, 20, 5)
    splan.close()
    salg = frames * (W * H * 2 + ow * oh * 3 * 2)
    interp = {"workload": "Cr2sRawInterpolator 4:2:0 v2 -> %dx%d RGB, %d frames/step" % (ow, oh, frames),
              "ms_per_step": round(sdt * 1e3, 4),
              "mpix_per_s": round(frames * ow * oh / sdt / 1e6, 1)}
    if skt:
        interp.update(kernel=skt[0], avg_kernel_ms=round(skt[1], 5),
                      achieved_gbps=round(salg / (skt[1] * 1e-3) / 1e9, 1),
                      frac_of_8tbps=round(salg / (skt[1] * 1e-3) / 8e12, 4))
    alg = frames * (scan_len + W * H * 2)
    return {
        "interpolate": interp,
        "workload": "Cr2Decompressor <3,2,2> (sRaw1) 3960x2640 px = %dx%d samples, 3 slices, "
                    "%d frames/step" % (W, H, frames),
        "msamples_per_s": round(frames * W * H / dt / 1e6, 1),
        "mpix_per_s": round(frames * 3960 * 2640 / dt / 1e6, 1),
        "ms_per_step": round(dt * 1e3, 4),
        "bit_exact": exact,
        "entropy_bits_per_sample": round(scan_len * 8 / (W * H), 3),
        "algorithmic_bytes_per_step": alg,
        "achieved_gbps_whole_pipeline": round(alg / dt / 1e9, 1),
    }


def run_cfg4(ctx, torch, log, steps=10, warmup=2):
    """configs[3]: 8192x5464 as 2x2 DNG tiles of 4096x2732 (one plan, tiles-parallel)."""
    from rawspeed_amd import abi, synth
    W, H, tw, th = 8192, 5464, 4096, 2732
    src = synth.sensor_image(W, H, 14, seed=2)
    jobs, blobs, off, lens = [], [], 0, []
    for ty in range(2):
        for tx in range(2):
            d, data, scan_len, bits = make_tile(src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw],
                                                tx * tw, ty * th, tw, th)
            j = abi.LJpegJob()
            j.desc = d
            j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
            j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
                out_pitch(W), W, H, 1, 1
            jobs.append(j)
            blobs.append(data)
            lens.append(scan_len)
            off += data.size
    inp = torch.from_numpy(np.concatenate(blobs)).cuda()
    out = torch.zeros(out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    got = out.cpu().numpy().view(np.uint16).reshape(H, out_pitch(W) // 2)[:, :W]
    exact = bool(np.array_equal(got, src)) and cons == lens
    alg = sum(lens) + W * H * 2
    res = {
        "workload": "LJpegDecompressor via DNG tiles: 8192x5464 as 2x2 tiles, 1 frame/step",
        "mpix_per_s": round(W * H / dt / 1e6, 1),
        "ms_per_step": round(dt * 1e3, 4),
        "bit_exact": exact,
        "entropy_bits_per_px": round(sum(lens) * 8 / (W * H), 3),
        "algorithmic_bytes_per_step": alg,
        "achieved_gbps_whole_pipeline": round(alg / dt / 1e9, 1),
        "frac_of_hbm_peak": round(alg / dt / 1e9 / 8000.0, 4),
    }
    _dominant(res, kt)
    return res


def make_cfg5_plan(ctx, torch, frames, distinct=4, seed0=1000):
    """configs[4]: a batch of independent 8192x5464 LJPEG frames (SOF3: 4096 x 5464,
    2 components, one scan each), `frames` per GPU; `distinct` different frames are
    synthesised and repeated.  Returns (plan, inp, out, meta)."""
    from rawspeed_amd import abi
    W, H = 8192, 5464
    blobs, srcs, lens = [], [], []
    from rawspeed_amd import synth
    for k in range(distinct):
        src = synth.sensor_image(W, H, 14, seed=seed0 + k)
        d, data, scan_len, bits = make_tile(src, 0, 0, W, H)
        blobs.append((d, data))
        lens.append(scan_len)
        if k == 0:
            srcs.append(src)
    jobs, off, parts = [], 0, []
    for f in range(frames):
        d, data = blobs[f % distinct]
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes = off, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
        parts.append(data)
        off += data.size
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
    meta = dict(W=W, H=H, src0=srcs[0], lens=[lens[f % distinct] for f in range(frames)],
                alg_bytes=sum(lens[f % distinct] for f in range(frames)) + frames * W * H * 2,
                bits_per_px=sum(lens) * 8 / (distinct * W * H))
    return plan, inp, out, meta


def cpu_baseline_cr2(d, data, W, H, budget_s=10.0):
    from oracle_lib import Ref
    if not Ref.available():
        return None
    ref = Ref()
    img = ref.image(W, H, 1)
    st, _ = ref.cr2(d, data, img)
    assert st == 0
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 5 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        ref.cr2(d, data, img)
        times.append(time.perf_counter() - t0)
    return {"value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
            "kind": "reference",
            "sample": "Cr2Decompressor::decompress of the unmodified reference on the same "
                      "6720x4480 stream, 1 thread (the decoder has no internal threading), "
                      "best of %d" % len(times)}


def run_nikon(ctx, torch, log, frames=8, steps=10, warmup=2, cpu=True):
    """NikonDecompressor: 14-bit lossless NEF of a 6016x4016 sensor
    (tree 5, identity curve), default output mode = curve table with dither."""
    import nikon_cases as N
    from rawspeed_amd import abi, synth
    W, H, bits = 6016, 4016, 14
    rng = np.random.default_rng(8)
    src = synth.sensor_image(W, H, 14, seed=8)
    meta = N.metadata(70, 0, [2000, 2500, 2500, 3000])
    P = N.parse(meta, bits, H)
    pu = P["p_up"]
    data, sym_bits = synth.nikon_encode(src, [pu[0][0], pu[0][1], pu[1][0], pu[1][1]],
                                        synth.NIKON_TREE[P["huff_select"]])
    data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
    out = {"workload": "NikonDecompressor 14-bit lossless %dx%d, %d frames/step" % (W, H, frames),
           "entropy_bits_per_px": round(sym_bits / (W * H), 3)}
    for mode, unc in (("curve_dither", 0), ("uncorrected", 1)):
        d = N.desc(P, bits, bool(unc))
        jobs = []
        for f in range(frames):
            j = abi.NikonJob()
            j.desc = d
            j.in_offset, j.in_bytes = f * data.size, data.size
            j.img_offset = f * out_pitch(W) * H
            j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
                out_pitch(W), W, H, 1, 1
            jobs.append(j)
        inp = torch.from_numpy(np.tile(data, frames)).cuda()
        outb = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
        plan = ctx.nikon_plan(jobs)
        dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
        plan.close()
        got = outb[-out_pitch(W) * H:].cpu().numpy().view(np.uint16).reshape(
            H, out_pitch(W) // 2)[:, :W]
        r = {"mpix_per_s": round(frames * W * H / dt / 1e6, 1),
             "ms_per_step": round(dt * 1e3, 4)}
        if unc:
            r["bit_exact"] = bool(np.array_equal(got, src))
        else:
            dith = got.copy()  # checked against the reference build below
        out[mode] = r
        log("nikon %s: %s" % (mode, r))
        del inp, outb
    if cpu:
        try:
            from oracle_lib import Ref
            if Ref.available():
                ref = Ref()
                img = ref.image(W, H, 1)
                assert ref.nikon(meta, bits, data, img, False) == 0
                out["curve_dither"]["bit_exact"] = bool(np.array_equal(img.pixels(), dith))
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    ref.nikon(meta, bits, data, img, False)
                    times.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {
                    "value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
                    "kind": "reference",
                    "sample": "NikonDecompressor::decompress of the unmodified reference on the "
                              "same stream (curve + dither), 1 thread, best of 3"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def run_hasselblad(ctx, torch, log, frames=4, steps=10, warmup=2, cpu=True):
    """HasselbladDecompressor: 8272x6200 16-bit frames (H5D-50c class),
    pair-coded symbols on an MSB32 stream."""
    import cases
    from rawspeed_amd import abi, synth
    W, H = 8272, 6200
    src = synth.sensor_image(W, H, 14, seed=11)
    data, sym_bits = synth.hasselblad_encode(src, 0x2000, cases.FULL17)
    data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
    d = abi.HasselbladDesc.make(cases.FULL17, 0x2000)
    jobs = []
    for f in range(frames):
        j = abi.HasselbladJob()
        j.desc = d
        j.in_offset, j.in_bytes = f * data.size, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    outb = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.hasselblad_plan(jobs)
    dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
    plan.close()
    got = outb[-out_pitch(W) * H:].cpu().numpy().view(np.uint16).reshape(
        H, out_pitch(W) // 2)[:, :W]
    out = {"workload": "HasselbladDecompressor %dx%d, %d frames/step" % (W, H, frames),
           "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
           "ms_per_step": round(dt * 1e3, 4),
           "bit_exact": bool(np.array_equal(got, src)),
           "entropy_bits_per_px": round(sym_bits / (W * H), 3)}
    if cpu:
        try:
            from oracle_lib import Ref
            if Ref.available():
                ref = Ref()
                img = ref.image(W, H, 1)
                assert ref.hasselblad(d, data, img)[0] == 0
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    ref.hasselblad(d, data, img)
                    times.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {
                    "value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
                    "kind": "reference",
                    "sample": "HasselbladDecompressor::decompress of the unmodified reference on "
                              "the same stream, 1 thread, best of 3"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def run_sony_arw1(ctx, torch, log, frames=8, steps=10, warmup=2, cpu=True):
    """SonyArw1Decompressor: the A100's 3881x2608 12-bit frame
    (ArwDecoder.cpp:128-129), column-major stream, one running predictor."""
    from rawspeed_amd import abi, synth
    W, H = 3881, 2608
    src = (synth.sensor_image(W + 1, H, 14, seed=12)[:, :W] >> 2).astype(np.uint16)
    data, sym_bits = synth.sony_arw1_encode(src)
    data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
    jobs = []
    for f in range(frames):
        j = abi.SonyArw1Job()
        j.in_offset, j.in_bytes = f * data.size, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    outb = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.sony_arw1_plan(jobs)
    dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
    plan.close()
    got = outb[-out_pitch(W) * H:].cpu().numpy().view(np.uint16).reshape(
        H, out_pitch(W) // 2)[:, :W]
    out = {"workload": "SonyArw1Decompressor %dx%d, %d frames/step" % (W, H, frames),
           "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
           "ms_per_step": round(dt * 1e3, 4),
           "bit_exact": bool(np.array_equal(got, src)),
           "entropy_bits_per_px": round(sym_bits / (W * H), 3)}
    if cpu:
        try:
            from oracle_lib import Ref
            if Ref.available():
                ref = Ref()
                img = ref.image(W, H, 1)
                assert ref.sony_arw1(data, img) == 0
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    ref.sony_arw1(data, img)
                    times.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {
                    "value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
                    "kind": "reference",
                    "sample": "SonyArw1Decompressor::decompress of the unmodified reference on "
                              "the same stream, 1 thread, best of 3"}
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def run_variants(ctx, torch, log, frames=8, steps=50, warmup=20):
    """The fixed-layout UncompressedDecompressor entry points at the
    cfg2 sensor size: decode12BitRawWithControl<big>, decode12BitRawUnpacked-
    LeftAligned<little>, decode8BitRaw<true>; same timing as the headline."""
    from rawspeed_amd import abi
    W, H = 8280, 5520
    out = {}
    rng = np.random.default_rng(9)
    for name, variant, big in (("12bit_with_control_be", 1, 1),
                               ("12bit_left_aligned_le", 2, 0), ("8bit_raw", 0, 0)):
        bpl = (W, 12 * W // 8 + (W + 2) // 10, 2 * W)[variant]
        in_stride = (bpl * H + 15) // 16 * 16
        pitch = out_pitch(W)
        frame = rng.integers(0, 256, size=in_stride, dtype=np.uint8)
        inp = torch.from_numpy(np.tile(frame, frames)).cuda()
        outb = torch.empty(frames * pitch * H, dtype=torch.uint8, device="cuda")
        jobs = []
        for f in range(frames):
            j = abi.UnpackVariantJob()
            j.desc = abi.UnpackVariantDesc(variant, big, W, H)
            j.in_offset, j.in_bytes, j.img_offset = f * in_stride, bpl * H, f * pitch * H
            j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = pitch, W, H, 1, 1
            jobs.append(j)
        plan = ctx.unpack_variant_plan(jobs)
        dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
        plan.close()
        alg = frames * H * (bpl + 2 * W)
        r = {"frames": frames, "ms_per_step": round(dt * 1e3, 4),
             "gpix_per_s": round(frames * W * H / dt / 1e9, 1)}
        if kt:
            r["kernel"] = kt[0]
            r["avg_kernel_ms"] = round(kt[1], 5)
            r["achieved_gbps"] = round(alg / (kt[1] * 1e-3) / 1e9, 1)
            r["frac_of_8tbps"] = round(alg / (kt[1] * 1e-3) / 8e12, 4)
        out[name] = r
        log("variant %s: %s" % (name, r))
        del inp, outb
    return out


def run(ctx, torch, log):
    out = {}
    try:
        out["uncompressed_variants_8280x5520"] = run_variants(ctx, torch, log)
    except Exception as e:
        out["uncompressed_variants_8280x5520"] = {"error": repr(e)}
    r3, ref_args = run_cfg3(ctx, torch, log)
    out["cfg3_cr2_6720x4480"] = r3
    out["cfg4_dng_tiles_8192x5464"] = run_cfg4(ctx, torch, log)
    try:
        out["nikon_lossless14_6016x4016"] = run_nikon(ctx, torch, log)
    except Exception as e:
        out["nikon_lossless14_6016x4016"] = {"error": repr(e)}
    try:
        out["hasselblad_8272x6200"] = run_hasselblad(ctx, torch, log)
    except Exception as e:
        out["hasselblad_8272x6200"] = {"error": repr(e)}
    try:
        out["sony_arw1_3881x2608"] = run_sony_arw1(ctx, torch, log)
    except Exception as e:
        out["sony_arw1_3881x2608"] = {"error": repr(e)}
    try:
        out["cr2_sraw1_3960x2640"] = run_sraw(ctx, torch, log)
    except Exception as e:
        out["cr2_sraw1_3960x2640"] = {"error": repr(e)}
    try:
        out["cfg3_cpu_baseline"] = cpu_baseline_cr2(*ref_args)
    except Exception as e:
        out["cfg3_cpu_baseline"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    import argparse
    import json
    import torch
    import __graft_entry__ as ge
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    ge.build()
    from rawspeed_amd import capi
    ctx = capi.Context(0)
    if args.only == "cfg3":
        r, _ = run_cfg3(ctx, torch, print, frames=args.frames, steps=args.steps)
        print(json.dumps(r, indent=1))
    elif args.only == "variants":
        print(json.dumps(run_variants(ctx, torch, print, frames=args.frames,
                                      steps=args.steps), indent=1))
    elif args.only == "nikon":
        print(json.dumps(run_nikon(ctx, torch, print, frames=args.frames, steps=args.steps),
                         indent=1))
    elif args.only == "sony":
        print(json.dumps(run_sony_arw1(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "hasselblad":
        print(json.dumps(run_hasselblad(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "sraw":
        print(json.dumps(run_sraw(ctx, torch, print, frames=args.frames, steps=args.steps),
                         indent=1))
    elif args.only == "cfg4":
        print(json.dumps(run_cfg4(ctx, torch, print, steps=args.steps), indent=1))
    else:
        print(json.dumps(run(ctx, torch, print), indent=1))
