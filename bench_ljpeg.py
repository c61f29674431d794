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


def _alt():
    from rawspeed_amd import synth
    return (synth.ALT_COUNTS, synth.ALT_VALUES)


def make_tile(src_tile, tx, ty, tw, th, rows_per_ri=0, want_blob=False, two_tables=False):
    """One DNG-style tile: SOF3 (tw/2) x th, 2 components, predictor 1.  Returns the
    descriptor + entropy-coded scan for the C-ABI and (want_blob) the whole SOI..EOI
    container for AbstractDngDecompressor.  two_tables: a code of its own per component
    (DHT slots 0 and 1), as DNG writers emit them."""
    from rawspeed_amd import abi, synth
    slots = [0, 1] if two_tables else [0, 0]
    tabs = [_nikon(), _alt()] if two_tables else [_nikon()]
    blob, nh, scan_len, bits = synth.ljpeg_container(np.ascontiguousarray(src_tile), 2, 14,
                                                     slots, tabs,
                                                     rows_per_ri=rows_per_ri)
    d = abi.LJpegDesc()
    d.tile_x, d.tile_y, d.tile_w, d.tile_h = tx, ty, tw, th
    d.mcu_w, d.mcu_h, d.frame_w, d.frame_h = 2, 1, tw // 2, th
    d.n_comp, d.rows_per_restart_interval = 2, rows_per_ri if rows_per_ri else th
    abi.fill_recipe(d, synth.huff_tables(*tabs), slots, [1 << 13] * 2)
    pad = (-(scan_len + 2)) % 16 + 16
    data = np.concatenate([blob[nh:nh + scan_len + 2], np.zeros(pad, np.uint8)])
    if want_blob:
        return d, data, scan_len, bits, blob
    return d, data, scan_len, bits


def gpu_frame(out_t, f, W, H):
    """frame f of a batch output tensor as a (H, W) uint16 array"""
    op = out_pitch(W)
    return out_t[f * op * H:(f + 1) * op * H].cpu().numpy().view(np.uint16) \
        .reshape(H, op // 2)[:, :W]


def host_threads(ref):
    return max(1, min(os.cpu_count() or 1, ref.lib.ref_max_threads()))


def ref_scan_baseline(kind, descs, datas, W, H, what, budget_s=8.0):
    """The unmodified reference on this host: one frame on one thread
    (the decoders have no threading of their own) and N frames on N cores (one frame
    per OpenMP thread, the shape of rstest.cpp:570).  Returns (the decoded distinct
    frames, the cpu_baseline dict)."""
    from oracle_lib import Ref
    if not Ref.available():
        return None, None
    ref = Ref()
    n = len(descs)
    imgs = [ref.image(W, H, 1) for _ in range(n)]
    one = []
    for k in range(n):
        t0 = time.perf_counter()
        st, _ = (ref.ljpeg if kind == 0 else ref.cr2)(descs[k], datas[k], imgs[k])
        one.append(time.perf_counter() - t0)
        assert st == 0, ref.last_error()
    frames = [i.pixels().copy() for i in imgs]
    nt = host_threads(ref)
    pimgs = imgs + [ref.image(W, H, 1) for _ in range(nt - n)] if nt > n else imgs[:nt]
    pd = [descs[k % n] for k in range(len(pimgs))]
    pdata = [datas[k % n] for k in range(len(pimgs))]
    ref.scan_frames_parallel(pimgs, pd, pdata, kind, nt)  # page touch
    best, t_end = None, time.perf_counter() + budget_s
    reps = 0
    while reps < 3 or (reps < 5 and time.perf_counter() < t_end):
        t0 = time.perf_counter()
        st = ref.scan_frames_parallel(pimgs, pd, pdata, kind, nt)
        dt = time.perf_counter() - t0
        assert st == 0
        best = dt if best is None else min(best, dt)
        reps += 1
    cpu = {"value": round(len(pimgs) * W * H / best / 1e6, 1), "unit": "MPix/s",
           "cores": nt, "kind": "reference",
           "single_thread_value": round(W * H / min(one) / 1e6, 1),
           "sample": "%s of the unmodified reference (oracle/_ref): %d frames on %d "
                     "threads, one frame per thread (best of %d); single_thread_value = "
                     "one frame on one thread" % (what, len(pimgs), nt, reps)}
    return frames, cpu


def ref_all_threads(ref, W, H, call, what, budget_s=6.0, max_frames=256):
    """SURVEY 8(d)'s second CPU figure for a decompressor without a parallel entry point in
    the reference wrapper: one frame per host thread, all threads at once (the calls release
    the GIL: N Python threads = N busy cores).  `call(img)` decodes the leg's stream into
    `img`.  Returns the fields to merge into the leg's cpu_baseline."""
    from concurrent.futures import ThreadPoolExecutor
    nt = min(host_threads(ref), max_frames)
    imgs = [ref.image(W, H, 1) for _ in range(nt)]
    best, reps = None, 0
    with ThreadPoolExecutor(max_workers=nt) as ex:
        list(ex.map(call, imgs))          # page touch
        t_end = time.perf_counter() + budget_s
        while reps < 2 or (reps < 4 and time.perf_counter() < t_end):
            t0 = time.perf_counter()
            list(ex.map(call, imgs))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            reps += 1
    return {"value_all_threads": round(nt * W * H / best / 1e6, 1), "cores_all_threads": nt,
            "sample_all_threads": "%s: %d frames on %d threads, one frame per thread (best of %d)"
                                  % (what, nt, nt, reps)}


LAST_KERNEL_TABLE = None  # per-kernel averages (ms per run) of the last _time_plan call


def _time_plan(torch, plan, inp, out, steps, warmup, with_results=False):
    """(seconds per step with timing off, dominant kernel (name, ms, runs), consumed).
    with_results: every step also fetches the per-job results (a host round trip; it is
    there that a stream which did not converge in-stream is finished).
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
        if with_results:
            assert plan.results()[0] == 0
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


def _roofline(res, alg, dt, kt):
    """The leg's `roofline` object, measured IN THIS RUN: algorithmic bytes of a step (the
    entropy-coded / packed input read once + the 16-bit image written once) over (a) the
    whole step -- every kernel of the plan, wall clock over the timed steps -- and (b) the
    dominant kernel's average launch duration from the hipEvents the library records after
    every launch (rsx_plan_set_timing; the table is in `kernels_ms`)."""
    r = {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "algorithmic_bytes": int(alg),
         "step_ms": round(dt * 1e3, 4),
         "achieved": round(alg / dt / 1e9, 1), "frac": round(alg / dt / 1e9 / 8000.0, 4),
         "measured": "in this run: wall clock of the timed steps; hipEvents after every launch"}
    tab = res.get("kernels_ms") or LAST_KERNEL_TABLE
    if kt:
        r["kernel"] = kt[0]
        r["avg_kernel_ms"] = round(kt[1], 5)
        r["frac_if_only_that_kernel_ran"] = round(alg / (kt[1] * 1e-3) / 8e12, 4)
    if tab:
        r["sum_of_kernels_ms"] = round(sum(tab.values()), 4)
    res["roofline"] = r
    return res


def _lj_result(workload, frames, W, H, dt, scan_bytes, exact, kt, extra=None):
    alg = scan_bytes + frames * W * H * 2
    res = {
        "workload": workload,
        "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
        "ms_per_step": round(dt * 1e3, 4),
        "bit_exact": exact,
        "entropy_bits_per_px": round(scan_bytes * 8 / (frames * W * H), 3),
        "algorithmic_bytes_per_step": alg,
        "achieved_gbps_whole_pipeline": round(alg / dt / 1e9, 1),
        "frac_of_hbm_peak": round(alg / dt / 1e9 / 8000.0, 4),
    }
    if extra:
        res.update(extra)
    _dominant(res, kt)
    _roofline(res, alg, dt, kt)
    return res


def _cr2_batch(ctx, torch, frames_desc, W, H):
    from rawspeed_amd import abi
    jobs, off, parts = [], 0, []
    for f, (d, data) in enumerate(frames_desc):
        j = abi.Cr2Job()
        j.desc = d
        j.in_offset, j.in_bytes = off, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
        parts.append(data)
        off += data.size
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(len(frames_desc) * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    return ctx.cr2_plan(jobs), inp, out


def run_cfg3(ctx, torch, log, frames=8, steps=10, warmup=2, cpu=True):
    """configs[2]: CR2-style 6720x4480, 2 components, 3 slices; `frames` DIFFERENT
    frames per step (seeds 1..frames), every one compared with the reference build."""
    W, H = 6720, 4480
    made = [make_cr2_frame(W, H, (3, 2240, 2240), seed=1 + f) for f in range(frames)]
    plan, inp, out = _cr2_batch(ctx, torch, [(m[0], m[1]) for m in made], W, H)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    exact = all(c == m[3] for c, m in zip(cons, made))
    ref_frames, cpu_b = (None, None)
    if cpu:
        ref_frames, cpu_b = ref_scan_baseline(1, [m[0] for m in made], [m[1] for m in made],
                                              W, H, "Cr2Decompressor::decompress")
    checked = "source image"
    for f in range(frames):
        got = gpu_frame(out, f, W, H)
        exact = exact and bool(np.array_equal(got, made[f][2]))
        if ref_frames is not None:
            exact = exact and bool(np.array_equal(got, ref_frames[f]))
            checked = "oracle/_ref (every frame) and the source images"
    res = _lj_result("Cr2Decompressor <2,1,1> 6720x4480, 3 slices, %d different frames/step"
                     % frames, frames, W, H, dt, sum(m[3] for m in made), exact, kt,
                     {"bit_exact_against": checked})
    if cpu_b:
        res["cpu_baseline"] = cpu_b
    return res, None


def run_cfg3_uniform(ctx, torch, log, frames=4, steps=10, warmup=2):
    """SURVEY 8(d) second distribution: uniform-random 14-bit values (~21 bit/px with
    this table: the entropy-coded input is larger than the 16 bit/px output)."""
    from rawspeed_amd import abi, synth
    import cases
    W, H = 6720, 4480
    made = []
    for f in range(frames):
        src = synth.uniform(W * H, 14, 77 + f).reshape(H, W)
        rows = cases.cr2_stream_from_image(src, 2, W // 2, H, cases.cr2_slices(3, 2240, 2240))
        scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [_nikon(), _nikon()])
        d = abi.Cr2Desc()
        d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
        d.frame_w, d.frame_h = W // 2, H
        d.num_slices, d.slice_width, d.last_slice_width = 3, 2240, 2240
        abi.fill_recipe(d, synth.huff_tables(_nikon()), [0, 0], [1 << 13] * 2)
        pad = (-(len(scan) + 2)) % 16 + 16
        data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
        made.append((d, data, src, len(scan)))
    plan, inp, out = _cr2_batch(ctx, torch, [(m[0], m[1]) for m in made], W, H)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    exact = all(c == m[3] for c, m in zip(cons, made))
    for f in range(frames):
        exact = exact and bool(np.array_equal(gpu_frame(out, f, W, H), made[f][2]))
    return _lj_result("Cr2Decompressor <2,1,1> 6720x4480, uniform-random 14-bit values "
                      "(worst case for the entropy stage), %d frames/step" % frames,
                      frames, W, H, dt, sum(m[3] for m in made), exact, kt)


def clipped_image(W, H, seed):
    """A frame as cameras deliver them: sensor-like, ~10 % of it blown out (a disc of
    16383) and a black border (masked pixels) -- constant regions make the bit stream
    periodic, which a self-synchronising decoder has to cope with."""
    from rawspeed_amd import synth
    src = synth.sensor_image(W, H, 14, seed=seed)
    yy, xx = np.ogrid[:H, :W]
    r2 = 0.10 * W * H / np.pi
    src[(yy - H * 0.45) ** 2 + (xx - W * 0.55) ** 2 < r2] = 16383
    b = 48
    src[:b, :] = 0
    src[-b:, :] = 0
    src[:, :b] = 0
    src[:, -b:] = 0
    return src


def run_clipped(ctx, torch, log, frames=8, steps=10, warmup=2):
    """cfg-3 shape with blown highlights and a black border, against the reference."""
    from rawspeed_amd import abi, synth
    import cases
    W, H = 6720, 4480
    made = []
    for f in range(frames):
        src = clipped_image(W, H, 31 + f)
        rows = cases.cr2_stream_from_image(src, 2, W // 2, H, cases.cr2_slices(3, 2240, 2240))
        scan, bits = synth.ljpeg_encode_scan(rows, 2, [1 << 13] * 2, [_nikon(), _nikon()])
        d = abi.Cr2Desc()
        d.n_comp, d.x_s_f, d.y_s_f = 2, 1, 1
        d.frame_w, d.frame_h = W // 2, H
        d.num_slices, d.slice_width, d.last_slice_width = 3, 2240, 2240
        abi.fill_recipe(d, synth.huff_tables(_nikon()), [0, 0], [1 << 13] * 2)
        pad = (-(len(scan) + 2)) % 16 + 16
        data = np.concatenate([scan, np.array([0xFF, 0xD9], np.uint8), np.zeros(pad, np.uint8)])
        made.append((d, data, src, len(scan)))
    plan, inp, out = _cr2_batch(ctx, torch, [(m[0], m[1]) for m in made], W, H)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    ktab = dict(LAST_KERNEL_TABLE or {})
    # once more with the per-job results fetched every step: nothing may be left to the
    # host-side fallback unseen (the fetch costs a synchronisation per step)
    dt_r, _, _ = _time_plan(torch, plan, inp, out, steps, warmup, with_results=True)
    exact = all(c == m[3] for c, m in zip(cons, made))
    ref_frames, _ = ref_scan_baseline(1, [made[0][0]], [made[0][1]], W, H, "Cr2Decompressor")
    for f in range(frames):
        exact = exact and bool(np.array_equal(gpu_frame(out, f, W, H), made[f][2]))
    if ref_frames is not None:
        exact = exact and bool(np.array_equal(gpu_frame(out, 0, W, H), ref_frames[0]))
    res = _lj_result("Cr2Decompressor <2,1,1> 6720x4480 with ~10 %% blown highlights (16383) "
                     "and a 48-px black border, %d different frames/step" % frames,
                     frames, W, H, dt, sum(m[3] for m in made), exact, kt,
                     {"ms_per_step_with_results_fetch": round(dt_r * 1e3, 4)})
    res["kernels_ms"] = ktab
    res["roofline"]["sum_of_kernels_ms"] = round(sum(ktab.values()), 4)
    return res


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
    ktab_dec = dict(LAST_KERNEL_TABLE or {})
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
    if skt:
        interp["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                              "algorithmic_bytes": salg, "kernel": skt[0],
                              "avg_kernel_ms": round(skt[1], 5),
                              "achieved": round(salg / (skt[1] * 1e-3) / 1e9, 1),
                              "frac": round(salg / (skt[1] * 1e-3) / 8e12, 4),
                              "measured": "in this run: hipEvents around the launch"}
    alg = frames * (scan_len + W * H * 2)
    return _roofline({
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
        "kernels_ms": ktab_dec,
    }, alg, dt, kt)


def _dng_tiles(W, H, tw, th, seed, rows_per_ri=0, two_tables=False):
    """A WxH sensor-like image as DNG tiles of tw x th (right / bottom tiles overhang when
    W, H are not multiples: the part outside the image is padding, decodeRowN drops it)."""
    from rawspeed_amd import abi, synth
    src = synth.sensor_image(W, H, 14, seed=seed)
    jobs, datas, blobs, lens, off = [], [], [], [], 0
    for ty in range((H + th - 1) // th):
        for tx in range((W + tw - 1) // tw):
            part = src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile = np.full((th, tw), 1000, np.uint16)
            tile[:part.shape[0], :part.shape[1]] = part
            d, data, scan_len, bits, blob = make_tile(tile, tx * tw, ty * th, tw, th,
                                                      rows_per_ri, want_blob=True,
                                                      two_tables=two_tables)
            # AbstractDngDecompressor.cpp:64-68: the tile is clipped to the image
            d.tile_w = min(tw, W - tx * tw)
            d.tile_h = min(th, H - ty * th)
            j = abi.LJpegJob()
            j.desc = d
            j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
            j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
                out_pitch(W), W, H, 1, 1
            jobs.append(j)
            datas.append(data)
            blobs.append(blob)
            lens.append(scan_len)
            off += data.size
    return src, jobs, datas, blobs, lens


def _cfg4_variant(ctx, torch, W, H, tw, th, seed, rows_per_ri, steps, warmup, what, cpu,
                  two_tables=False):
    from oracle_lib import Ref
    src, jobs, datas, blobs, lens = _dng_tiles(W, H, tw, th, seed, rows_per_ri, two_tables)
    inp = torch.from_numpy(np.concatenate(datas)).cuda()
    out = torch.zeros(out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    got = gpu_frame(out, 0, W, H)
    exact = bool(np.array_equal(got, src))
    extra = {"bit_exact_against": "the source image"}
    if Ref.available():
        # the same tiles through AbstractDngDecompressor::decompress() of the unmodified
        # reference, its OpenMP fan-out over the tiles on all cores (only 4 are busy)
        ref = Ref()
        img = ref.image(W, H, 1)
        nt = host_threads(ref)
        st = ref.dng(img, 7, tw, th, blobs, threads=nt)
        exact = exact and st == 0 and bool(np.array_equal(got, img.pixels()))
        extra["bit_exact_against"] = "oracle/_ref (AbstractDngDecompressor::decompress) " \
                                     "and the source image"
        if cpu:
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                ref.dng(img, 7, tw, th, blobs, threads=nt)
                ts.append(time.perf_counter() - t0)
            t1 = []
            for _ in range(2):
                t0 = time.perf_counter()
                ref.dng(img, 7, tw, th, blobs, threads=1)
                t1.append(time.perf_counter() - t0)
            # (an OpenMP team of every host thread for four tiles costs the reference more
            # than the decode: its fairest figure is one thread per tile)
            t4 = []
            for _ in range(3):
                t0 = time.perf_counter()
                ref.dng(img, 7, tw, th, blobs, threads=min(nt, len(blobs)))
                t4.append(time.perf_counter() - t0)
            extra["cpu_baseline"] = {
                "value": round(W * H / min(ts) / 1e6, 1), "unit": "MPix/s", "cores": nt,
                "value_threads4": round(W * H / min(t4) / 1e6, 1),
                "kind": "reference", "single_thread_value": round(W * H / min(t1) / 1e6, 1),
                "sample": "AbstractDngDecompressor::decompress() of the unmodified reference "
                          "on the same %d tiles, OpenMP threads = %d (at most %d busy), best "
                          "of 5; single_thread_value = 1 thread" % (len(blobs), nt, len(blobs))}
    # consumed: full-height tiles end on their marker; bottom-overhanging ones stop early
    if H % th == 0:
        exact = exact and cons == lens
    return _lj_result(what, 1, W, H, dt, sum(lens), exact, kt, extra)


def run_cfg4(ctx, torch, log, steps=10, warmup=2, cpu=True, variants=True):
    """configs[3]: 8192x5464 as 2x2 DNG tiles of 4096x2732 (one plan, tiles-parallel),
    plus SURVEY 8(d)'s parity extras at full size: 8189x5462 (right / bottom tiles
    overhang, odd width) and the restart-interval variant."""
    res = _cfg4_variant(ctx, torch, 8192, 5464, 4096, 2732, 2, 0, steps, warmup,
                        "LJpegDecompressor via DNG tiles: 8192x5464 as 2x2 tiles, 1 frame/step",
                        cpu)
    if variants:
        res["two_tables"] = _cfg4_variant(
            ctx, torch, 8192, 5464, 4096, 2732, 2, 0, steps, warmup,
            "8192x5464 as 2x2 tiles, a Huffman table of its own per component (DHT slots 0, 1)",
            False, two_tables=True)
        res["two_tables_256x256_tiles"] = _cfg4_variant(
            ctx, torch, 8192, 5464, 256, 256, 2, 0, steps, warmup,
            "8192x5464 as 32x22 tiles of 256x256 (what Adobe's DNG converter writes), a Huffman "
            "table of its own per component", False, two_tables=True)
        res["overhang_8189x5462"] = _cfg4_variant(
            ctx, torch, 8189, 5462, 4096, 2732, 3, 0, steps, warmup,
            "8189x5462 as 2x2 tiles of 4096x2732 (overhanging right/bottom tiles)", False)
        res["restart_intervals"] = _cfg4_variant(
            ctx, torch, 8192, 5464, 4096, 2732, 2, 28, steps, warmup,
            "8192x5464 as 2x2 tiles, restart interval = 28 rows = 57344 MCUs (the DRI field is 16 bits), 98 intervals per tile", False)
    return res


def run_ljpeg3(ctx, torch, log, steps=10, warmup=2, n=3, tables_per_component=False):
    """A linear (3 components per pixel) 8192x5464 DNG as 2x2 LJPEG tiles of 4096x2732, MCU
    3 x 1 (LJpegDecompressor.cpp:102-105): with ONE Huffman table the single-pass kernel's <3>
    instantiation (round 5; until then the legacy route with its int16 difference scratch);
    with a table PER COMPONENT -- what DNG writers emit, AbstractLJpegDecoder.cpp:181-291 -- its
    table-per-phase instantiation (round 6).  n = 4: a 4-component scan over a 1-sample image
    (MCU 4 x 1), tables A B C D.  Every tile is compared with the image it was written from."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    from rawspeed_amd import abi
    cpp = 3 if n == 3 else 1
    W, H, tw, th = 8192, 5464, 4096, 2732
    pitch = (W * cpp * 2 + 15) // 16 * 16
    rng = np.random.default_rng(33)
    tables, index = (cases.NIKON,), None
    if tables_per_component:
        trng = np.random.default_rng(3303)
        tables = (cases.NIKON, cases.ALT) + tuple(
            cases.random_huffman_table(trng, n_cat=15, skew=1.5) for _ in range(n - 2))
        index = list(range(n))
    jobs, parts, tiles, off, scan_total = [], [], [], 0, 0
    for ty in range(2):
        for tx in range(2):
            d, data, tile_px, scan_len = cases.make_ljpeg_case(
                rng, img_w=W, img_h=H, cpp=cpp, tile=(tx * tw, ty * th, tw, th), mcu=(n, 1),
                tables=tables, table_index=index)
            pad = (-data.size) % 16
            j = abi.LJpegJob()
            j.desc = d
            j.in_offset, j.in_bytes, j.img_offset = off, data.size, 0
            j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
                pitch, W, H, cpp, int(cpp == 1)
            jobs.append(j)
            parts.append(np.concatenate([data, np.zeros(pad, np.uint8)]))
            tiles.append((tx, ty, tile_px, scan_len))
            off += data.size + pad
            scan_total += scan_len
    inp = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(pitch * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
    dt, kt, cons = _time_plan(torch, plan, inp, out, steps, warmup)
    got = out.cpu().numpy().view(np.uint16).reshape(H, pitch // 2)[:, :W * cpp]
    exact = list(cons) == [t[3] for t in tiles]
    for tx, ty, tile_px, _ in tiles:
        exact = exact and bool(np.array_equal(
            got[ty * th:(ty + 1) * th, tx * tw * cpp:(tx + 1) * tw * cpp], tile_px))
    plan.close()
    alg = scan_total + W * H * cpp * 2
    res = {"workload": "LJpegDecompressor, %d components (MCU %dx1), %s: %s 8192x5464 DNG as 2x2 "
                       "tiles of 4096x2732, 1 frame/step"
                       % (n, n, "a Huffman table per component (%s)" % " ".join("ABCD"[:n])
                          if tables_per_component else "one Huffman table",
                          "linear" if cpp == 3 else "1-sample"),
           "mpix_per_s": round(W * H / dt / 1e6, 1),
           "msamples_per_s": round(W * H * cpp / dt / 1e6, 1),
           "ms_per_step": round(dt * 1e3, 4), "bit_exact": exact,
           "bit_exact_against": "the image every tile was written from",
           "entropy_bits_per_sample": round(scan_total * 8 / (W * H * cpp), 3),
           "algorithmic_bytes_per_step": alg}
    _dominant(res, kt)
    _roofline(res, alg, dt, kt)
    return res


def cfg5_pick(g, distinct):
    """which of the `distinct` synthesised frames global frame g of the batch is (the
    rotation makes the shards of consecutive ranks differ)"""
    return (g + g // distinct) % distinct


def cfg5_frame_bytes(meta, g):
    return int(meta["blobs"][cfg5_pick(g, meta["distinct"])][1].size)


def cfg5_assemble(torch, meta, frames_global):
    """the packed input of the given global frames, on the device"""
    dev = meta.get("_dev")
    if dev is None:
        dev = meta["_dev"] = [torch.from_numpy(b[1]).cuda() for b in meta["blobs"]]
    return torch.cat([dev[cfg5_pick(g, meta["distinct"])] for g in frames_global])


def make_cfg5_plan(ctx, torch, frames, distinct=32, seed0=1000, first_frame=0):
    """configs[4]: a batch of independent 8192x5464 LJPEG frames (SOF3: 4096 x 5464,
    2 components, one scan each), `frames` on this GPU = global frames first_frame ..;
    `distinct` different frames (seeds seed0 + k) are synthesised, global frame g is
    number cfg5_pick(g).  Returns (plan, inp, out, meta)."""
    from rawspeed_amd import abi
    W, H = 8192, 5464
    distinct = max(1, min(distinct, 256))
    blobs, srcs, lens = [], [], []
    from concurrent.futures import ThreadPoolExecutor
    from rawspeed_amd import synth

    def one(k):  # (the stream writer is C behind ctypes: the threads run side by side)
        src = synth.sensor_image(W, H, 14, seed=seed0 + k)
        d, data, scan_len, bits = make_tile(src, 0, 0, W, H)
        return src, d, data, scan_len

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        for src, d, data, scan_len in ex.map(one, range(distinct)):
            blobs.append((d, data))
            lens.append(scan_len)
            srcs.append(src)
    pick = [cfg5_pick(first_frame + f, distinct) for f in range(frames)]
    jobs, off = [], 0
    for f in range(frames):
        d, data = blobs[pick[f]]
        j = abi.LJpegJob()
        j.desc = d
        j.in_offset, j.in_bytes = off, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
        off += data.size
    # the packed batch is assembled on the device (256 frames are 12 GB)
    meta = dict(W=W, H=H, srcs=srcs, distinct=distinct, blobs=blobs, pick=pick,
                lens=[lens[k] for k in pick],
                alg_bytes=sum(lens[k] for k in pick) + frames * W * H * 2,
                bits_per_px=sum(lens) * 8 / (distinct * W * H))
    inp = cfg5_assemble(torch, meta, range(first_frame, first_frame + frames))
    out = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.ljpeg_plan(jobs)
    return plan, inp, out, meta


def check_cfg5(out, meta, cons, frames, ref_frames=None):
    """every frame of the batch against its source image (and the reference's decode of
    it); compared on the device -- 256 frames are 23 GB"""
    import torch
    W, H, k = meta["W"], meta["H"], meta["distinct"]
    ok = list(cons) == list(meta["lens"])
    op = out_pitch(W)
    view = out.view(torch.int16).reshape(frames, H, op // 2)[:, :, :W]
    want = [torch.from_numpy(np.ascontiguousarray(s).view(np.int16)).cuda()
            for s in meta["srcs"]]
    if ref_frames is not None:
        for i in range(k):
            ok = ok and bool(np.array_equal(ref_frames[i], meta["srcs"][i]))
    for f in range(frames):
        ok = ok and bool(torch.equal(view[f], want[meta["pick"][f]]))
    return ok


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
             "ms_per_step": round(dt * 1e3, 4), "kernels_ms": dict(LAST_KERNEL_TABLE or {})}
        _roofline(r, frames * (data.size + W * H * 2), dt, kt)
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
                out["cpu_baseline"].update(ref_all_threads(
                    ref, W, H, lambda im: ref.nikon(meta, bits, data, im, False),
                    "NikonDecompressor::decompress"))
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
           "entropy_bits_per_px": round(sym_bits / (W * H), 3),
           "kernels_ms": dict(LAST_KERNEL_TABLE or {})}
    _roofline(out, frames * (data.size + W * H * 2), dt, kt)
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
                out["cpu_baseline"].update(ref_all_threads(
                    ref, W, H, lambda im: ref.hasselblad(d, data, im),
                    "HasselbladDecompressor::decompress", max_frames=64))
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
           "entropy_bits_per_px": round(sym_bits / (W * H), 3),
           "kernels_ms": dict(LAST_KERNEL_TABLE or {})}
    _roofline(out, frames * (data.size + W * H * 2), dt, kt)
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
                out["cpu_baseline"].update(ref_all_threads(
                    ref, W, H, lambda im: ref.sony_arw1(data, im),
                    "SonyArw1Decompressor::decompress"))
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def _prefix_leg(ctx, torch, log, what, W, H, frames, steps, warmup, plan_of, job_type, desc,
                data, src, sym_bits, ref_call, cpu):
    """Pentax / SamsungV1: the Nikon stream kind with its own table, predictors and range
    check (rsx_ljpeg_recon.hip); `frames` copies of one stream per step."""
    jobs = []
    for f in range(frames):
        j = job_type()
        j.desc = desc
        j.in_offset, j.in_bytes = f * data.size, data.size
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
    inp = torch.from_numpy(np.tile(data, frames)).cuda()
    outb = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = plan_of(jobs)
    dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
    plan.close()
    exact = all(bool(np.array_equal(gpu_frame(outb, f, W, H), src)) for f in (0, frames - 1))
    out = {"workload": "%s %dx%d, %d frames/step" % (what, W, H, frames),
           "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
           "ms_per_step": round(dt * 1e3, 4), "bit_exact": exact,
           "bit_exact_against": "the image the stream was written from",
           "entropy_bits_per_px": round(sym_bits / (W * H), 3),
           "kernels_ms": dict(LAST_KERNEL_TABLE or {})}
    _roofline(out, frames * (data.size + W * H * 2), dt, kt)
    if cpu:
        try:
            from oracle_lib import Ref
            if Ref.available():
                ref = Ref()
                img = ref.image(W, H, 1)
                assert ref_call(ref, img) == 0
                out["bit_exact"] = exact and bool(np.array_equal(img.pixels(), gpu_frame(outb, 0, W, H)))
                out["bit_exact_against"] = "the reference build's output and the source image"
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    ref_call(ref, img)
                    times.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {
                    "value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
                    "kind": "reference",
                    "sample": "%s::decompress of the unmodified reference on the same stream, "
                              "1 thread, best of 3" % what}
                out["cpu_baseline"].update(ref_all_threads(
                    ref, W, H, lambda im: ref_call(ref, im), "%s::decompress" % what))
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def run_pentax(ctx, torch, log, frames=8, steps=10, warmup=2, cpu=True):
    """PentaxDecompressor: a K-1-sized 7392x4950 14-bit frame, the makernote
    ("modern") code table."""
    import nikon_cases as N
    from rawspeed_amd import abi
    W, H = 7392, 4950
    src = N.smooth15(np.random.default_rng(41), H, W, maxv=16383, sigma=9.0)
    tree = N.PENTAX_MODERN
    data, sym_bits = N.pentax_encode(src, tree)
    data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
    meta = N.pentax_metadata(tree)
    return _prefix_leg(ctx, torch, log, "PentaxDecompressor", W, H, frames, steps, warmup,
                       ctx.pentax_plan, abi.PentaxJob, N.pentax_desc(tree), data, src, sym_bits,
                       lambda ref, img: ref.pentax(meta, data, img), cpu)


def run_samsung_v1(ctx, torch, log, frames=8, steps=10, warmup=2, cpu=True):
    """SamsungV1Decompressor: an NX-sized 5472x3648 12-bit frame."""
    import nikon_cases as N
    from rawspeed_amd import abi, synth
    W, H = 5472, 3648
    src = N.smooth15(np.random.default_rng(43), H, W, maxv=4095, sigma=6.0)
    data, sym_bits = synth.prefix_encode(src, [0, 0, 0, 0], synth.SAMSUNG_V1_TAB)
    data = np.concatenate([data, np.zeros(16 + (-len(data)) % 16, np.uint8)])
    return _prefix_leg(ctx, torch, log, "SamsungV1Decompressor", W, H, frames, steps, warmup,
                       ctx.samsung_v1_plan, abi.SamsungV1Job,
                       abi.SamsungV1Desc.make(synth.SAMSUNG_V1_TAB), data, src, sym_bits,
                       lambda ref, img: ref.samsung_v1(12, data, img), cpu)


def make_samsung_v2_frame(W, H, bits=14, seed=21, band=34):
    """A WxH SamsungV2 stream (header + rows).  The Python writer (tests/samsung_v2_cases.py)
    does 0.1 MPix/s, so `band` rows are written for a sensor-like target and the rows
    behind the first two are repeated down the frame: every row of such a stream is a valid
    row at any position of its parity (the first two rows are special, :172-173, :325-326),
    the decoders reconstruct SOME image from it -- the same one, which is what is checked."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import samsung_v2_cases as V2
    from rawspeed_amd import synth
    rng = np.random.default_rng(seed)
    band += band & 1
    target = (synth.sensor_image(W, band, 14, seed=seed) >> (14 - bits)).astype(np.int64)
    rows = []
    data, _ = V2.encode(rng, target, bits, 0, rows_out=rows)
    # the header with the frame's height: the writer's, re-written
    head = bytearray(data[:16].tobytes())
    bits_s = "".join(format(int.from_bytes(head[4 * k:4 * k + 4], "little"), "032b") for k in range(4))
    bits_s = bits_s[:48] + format(H, "016b") + bits_s[64:]
    head = b"".join(int(bits_s[32 * k:32 * k + 32], 2).to_bytes(4, "little") for k in range(4))
    out = [np.frombuffer(head, np.uint8)] + rows[:2]
    k = 2
    for r in range(2, H):
        out.append(rows[k])
        k = k + 1 if k + 1 < band else 2
    return np.concatenate(out + [np.zeros(16, np.uint8)])


def run_samsung_v2(ctx, torch, log, frames=4, steps=5, warmup=1, cpu=True):
    """SamsungV2Decompressor: an NX1-sized 6480x4320 14-bit frame."""
    from rawspeed_amd import abi
    W, H, bits = 6480, 4320, 14
    data = make_samsung_v2_frame(W, H, bits)
    d, _ = abi.SamsungV2Desc.from_header(data[:16])
    payload = data[16:]
    payload = np.concatenate([payload, np.zeros((-payload.size) % 16, np.uint8)])
    jobs = []
    for f in range(frames):
        j = abi.SamsungV2Job()
        j.desc = d
        j.in_offset, j.in_bytes = f * payload.size, data.size - 16
        j.img_offset = f * out_pitch(W) * H
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            out_pitch(W), W, H, 1, 1
        jobs.append(j)
    inp = torch.from_numpy(np.tile(payload, frames)).cuda()
    outb = torch.zeros(frames * out_pitch(W) * H, dtype=torch.uint8, device="cuda")
    plan = ctx.samsung_v2_plan(jobs)
    dt, kt, _ = _time_plan(torch, plan, inp, outb, steps, warmup)
    ktab = dict(LAST_KERNEL_TABLE or {})
    plan.close()
    out = {"workload": "SamsungV2Decompressor %dx%d %d-bit, %d frames/step" % (W, H, bits, frames),
           "mpix_per_s": round(frames * W * H / dt / 1e6, 1),
           "ms_per_step": round(dt * 1e3, 4),
           "compressed_bits_per_px": round(data.size * 8 / (W * H), 3),
           "kernels_ms": ktab}
    _roofline(out, frames * (data.size + W * H * 2), dt, kt)
    if cpu:
        try:
            from oracle_lib import Ref
            if Ref.available():
                ref = Ref()
                img = ref.image(W, H, 1)
                assert ref.samsung_v2(bits, data, img) == 0
                want = img.pixels().copy()
                exact = True
                for f in (0, frames - 1):
                    exact = exact and bool(np.array_equal(gpu_frame(outb, f, W, H), want))
                out["bit_exact"] = exact
                out["checked_against"] = "the reference build's output"
                times = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    ref.samsung_v2(bits, data, img)
                    times.append(time.perf_counter() - t0)
                out["cpu_baseline"] = {
                    "value": round(W * H / min(times) / 1e6, 1), "unit": "MPix/s", "cores": 1,
                    "kind": "reference",
                    "sample": "SamsungV2Decompressor::decompress of the unmodified reference on "
                              "the same stream, 1 thread, best of 2"}
                # (the reference cannot use a second core for a frame, but it can for a batch:
                # what the device's 6 GPix/s is up against on this host)
                out["cpu_baseline"].update(ref_all_threads(
                    ref, W, H, lambda im: ref.samsung_v2(bits, data, im),
                    "SamsungV2Decompressor::decompress"))
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


def run_host_path(torch, log, reps=7):
    """What a patched rawspeed sees: the reference's own entry points of
    the GPU-backed build (oracle/_ref/librawspeed_rsx.so) on pageable host buffers --
    staging over PCIe included -- next to this box's plain pageable copy of the same
    bytes (hipMemcpy H2D of the input + D2H of the output, nothing else)."""
    from oracle_lib import Ref
    from rawspeed_amd import abi, synth
    here = os.path.join(ROOT, "oracle", "_ref", "librawspeed_rsx.so")
    if not os.path.exists(here):
        return {"error": "oracle/_ref/librawspeed_rsx.so is not built"}
    rsx = Ref(here)
    ref = Ref() if Ref.available() else None

    def copy_rate(n_in, n_out):
        hin = np.zeros(n_in, np.uint8)
        hout = np.zeros(n_out, np.uint8)
        din = torch.empty(n_in, dtype=torch.uint8, device="cuda")
        dout = torch.empty(n_out, dtype=torch.uint8, device="cuda")
        tin, tout = torch.from_numpy(hin), torch.from_numpy(hout)
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            din.copy_(tin)
            tout.copy_(dout)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def timed(fn):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            st = fn()
            dt = time.perf_counter() - t0
            assert st == 0, rsx.last_error()
            best = dt if best is None else min(best, dt)
        return best

    def leg(what, W, H, n_in, fn, check):
        n_out = W * H * 2
        fn()  # first call: context, module load
        dt = timed(fn)
        dc = copy_rate(n_in, n_out)
        return {"workload": what, "ms_per_call": round(dt * 1e3, 3),
                "mpix_per_s": round(W * H / dt / 1e6, 1),
                "link_gbps": round((n_in + n_out) / dt / 1e9, 1),
                "pageable_copy_ms": round(dc * 1e3, 3),
                "pageable_copy_gbps": round((n_in + n_out) / dc / 1e9, 1),
                "frac_of_pageable_copy_rate": round(dc / dt, 3),
                "bit_exact": bool(check())}

    out = {}
    # cfg 2: UncompressedDecompressor::readUncompressedRaw
    W, H, bps = 8192, 5464, 14
    px = synth.uniform(W * H, bps, 7).reshape(H, W)
    packed = synth.pack_rows(px, bps, 1)
    d = abi.UnpackDesc(0, 0, W, H, W * bps // 8, bps, 1)
    img = rsx.image(W, H, 1)
    out["cfg2_unpack_14bit_8192x5464"] = leg(
        "UncompressedDecompressor::readUncompressedRaw of the patched reference, one frame",
        W, H, packed.size, lambda: rsx.unpack(d, packed, img),
        lambda: np.array_equal(img.pixels(), px))
    # cfg 3: Cr2Decompressor::decompress
    W, H = 6720, 4480
    d3, data3, src3, n3, _ = make_cr2_frame(W, H, (3, 2240, 2240), seed=1)
    img3 = rsx.image(W, H, 1)
    out["cfg3_cr2_6720x4480"] = leg(
        "Cr2Decompressor::decompress of the patched reference, one frame",
        W, H, data3.size, lambda: rsx.cr2(d3, data3, img3)[0],
        lambda: np.array_equal(img3.pixels(), src3))
    # cfg 4: AbstractDngDecompressor::decompress -> one batched call for the 4 tiles
    W, H, tw, th = 8192, 5464, 4096, 2732
    src4, jobs, datas, blobs, lens = _dng_tiles(W, H, tw, th, 2)
    img4 = rsx.image(W, H, 1)
    # (rawspeed_get_number_of_processor_cores() = the tile count: an OpenMP team of all
    # 256 host threads for four tiles costs more than the decode)
    nt4 = min(host_threads(rsx), len(blobs))
    calls0 = rsx.rsx_host_calls()
    res4 = leg("AbstractDngDecompressor::decompress of the patched reference (4 LJPEG tiles, "
               "one batched call), one frame",
               W, H, sum(b.size for b in blobs), lambda: rsx.dng(img4, 7, tw, th, blobs, threads=nt4),
               lambda: np.array_equal(img4.pixels(), src4))
    res4["openmp_threads"] = nt4
    res4["rsx_calls_per_decompress"] = round((rsx.rsx_host_calls() - calls0) / (reps + 1), 2)
    out["cfg4_dng_tiles_8192x5464"] = res4
    # The same three calls with page-locked buffers (rsx.h: rsx_host_alloc / rsx_host_register,
    # INTEGRATION.md 6): the images come from the pool behind the patched AlignedAllocator
    # (what RawImageData::createData allocates from), the input arrays are registered in place
    # -- what an application does with its file buffer.
    try:
        from rawspeed_amd import capi
        if rsx.set_pinned_pool(True):
            cx = capi.Context(0)
            pinned = {}
            for key, W_, H_, data_, call, want in (
                    ("cfg2_unpack_14bit_8192x5464", 8192, 5464, packed,
                     lambda im: rsx.unpack(d, packed, im), px),
                    ("cfg3_cr2_6720x4480", 6720, 4480, data3,
                     lambda im: rsx.cr2(d3, data3, im)[0], src3),
                    ("cfg4_dng_tiles_8192x5464", 8192, 5464, None,
                     lambda im: rsx.dng(im, 7, tw, th, blobs, threads=nt4), src4)):
                arrays = [data_] if data_ is not None else list(blobs)
                reg = [a for a in arrays if cx.host_register(a) == 0]
                im = rsx.image(W_, H_, 1)   # (allocated with the pool on: page-locked)
                call(im)
                dt = timed(lambda: call(im))
                n_in = sum(a.size for a in arrays)
                pinned[key] = {"ms_per_call": round(dt * 1e3, 3),
                               "mpix_per_s": round(W_ * H_ / dt / 1e6, 1),
                               "link_gbps": round((n_in + W_ * H_ * 2) / dt / 1e9, 1),
                               "inputs_registered": len(reg) == len(arrays),
                               "bit_exact": bool(np.array_equal(im.pixels(), want)),
                               "ms_per_call_pageable": out[key]["ms_per_call"]}
                del im
                for a in reg:
                    cx.host_unregister(a)
            out["page_locked_buffers"] = pinned
            rsx.set_pinned_pool(False)
            cx.close()
    except Exception as e:
        out["page_locked_buffers"] = {"error": repr(e)}
    return out


def run(ctx, torch, log):
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # the other legs must survive
            out[name] = {"error": repr(e)}

    leg("uncompressed_variants_8280x5520", lambda: run_variants(ctx, torch, log))
    leg("cfg3_cr2_6720x4480", lambda: run_cfg3(ctx, torch, log)[0])
    leg("cfg3_uniform_random_14bit", lambda: run_cfg3_uniform(ctx, torch, log))
    leg("cfg3_clipped_highlights", lambda: run_clipped(ctx, torch, log))
    leg("cfg4_dng_tiles_8192x5464", lambda: run_cfg4(ctx, torch, log))
    leg("ljpeg_3comp_8192x5464", lambda: run_ljpeg3(ctx, torch, log))
    leg("ljpeg_3comp_3tables_8192x5464", lambda: run_ljpeg3(ctx, torch, log, tables_per_component=True))
    leg("ljpeg_4comp_1table_8192x5464", lambda: run_ljpeg3(ctx, torch, log, n=4))
    leg("ljpeg_4comp_4tables_8192x5464", lambda: run_ljpeg3(ctx, torch, log, n=4, tables_per_component=True))
    leg("nikon_lossless14_6016x4016", lambda: run_nikon(ctx, torch, log))
    leg("hasselblad_8272x6200", lambda: run_hasselblad(ctx, torch, log))
    leg("sony_arw1_3881x2608", lambda: run_sony_arw1(ctx, torch, log))
    leg("pentax_7392x4950", lambda: run_pentax(ctx, torch, log))
    leg("samsung_v1_5472x3648", lambda: run_samsung_v1(ctx, torch, log))
    leg("samsung_v2_6480x4320", lambda: run_samsung_v2(ctx, torch, log))
    leg("cr2_sraw1_3960x2640", lambda: run_sraw(ctx, torch, log))
    leg("host_path", lambda: run_host_path(torch, log))
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
    ap.add_argument("--no-cpu", action="store_true", help="skip the reference CPU baseline")
    args = ap.parse_args()
    ge.build()
    from rawspeed_amd import capi
    ctx = capi.Context(0)
    if args.only == "cfg3":
        r, _ = run_cfg3(ctx, torch, print, frames=args.frames, steps=args.steps,
                        cpu=not args.no_cpu)
        print(json.dumps(r, indent=1))
    elif args.only == "variants":
        print(json.dumps(run_variants(ctx, torch, print, frames=args.frames,
                                      steps=args.steps), indent=1))
    elif args.only == "nikon":
        print(json.dumps(run_nikon(ctx, torch, print, frames=args.frames, steps=args.steps),
                         indent=1))
    elif args.only == "sony":
        print(json.dumps(run_sony_arw1(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "ljpeg3":
        print(json.dumps(run_ljpeg3(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "ljpegpt":  # (a table per component: 3 and 4 components, next to one table)
        print(json.dumps({
            "ljpeg_3comp_8192x5464": run_ljpeg3(ctx, torch, print, steps=args.steps),
            "ljpeg_3comp_3tables_8192x5464": run_ljpeg3(ctx, torch, print, steps=args.steps,
                                                        tables_per_component=True),
            "ljpeg_4comp_1table_8192x5464": run_ljpeg3(ctx, torch, print, steps=args.steps, n=4),
            "ljpeg_4comp_4tables_8192x5464": run_ljpeg3(ctx, torch, print, steps=args.steps, n=4,
                                                        tables_per_component=True)}, indent=1))
    elif args.only == "pentax":
        print(json.dumps(run_pentax(ctx, torch, print, frames=args.frames, steps=args.steps,
                                    cpu=not args.no_cpu), indent=1))
    elif args.only == "samsung_v1":
        print(json.dumps(run_samsung_v1(ctx, torch, print, frames=args.frames, steps=args.steps,
                                        cpu=not args.no_cpu), indent=1))
    elif args.only == "samsung_v2":
        print(json.dumps(run_samsung_v2(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "hasselblad":
        print(json.dumps(run_hasselblad(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "sraw":
        print(json.dumps(run_sraw(ctx, torch, print, frames=args.frames, steps=args.steps),
                         indent=1))
    elif args.only == "uniform":
        print(json.dumps(run_cfg3_uniform(ctx, torch, print, steps=args.steps), indent=1))
    elif args.only == "clipped":
        print(json.dumps(run_clipped(ctx, torch, print, frames=args.frames, steps=args.steps),
                         indent=1))
    elif args.only == "host":
        print(json.dumps(run_host_path(torch, print), indent=1))
    elif args.only == "cfg4small":
        # what Adobe's DNG converter writes: 256 x 256 tiles, a table of its own per component
        print(json.dumps(_cfg4_variant(
            ctx, torch, 8192, 5464, 256, 256, 2, 0, args.steps, 2,
            "8192x5464 as 32x22 tiles of 256x256, a Huffman table of its own per component", False,
            two_tables=True), indent=1))
    elif args.only == "cfg4small1":
        print(json.dumps(_cfg4_variant(
            ctx, torch, 8192, 5464, 256, 256, 2, 0, args.steps, 2,
            "8192x5464 as 32x22 tiles of 256x256, one table", False), indent=1))
    elif args.only == "cfg4mt":
        print(json.dumps(_cfg4_variant(
            ctx, torch, 8192, 5464, 4096, 2732, 2, 0, args.steps, 2,
            "8192x5464 as 2x2 tiles, a Huffman table of its own per component", False,
            two_tables=True), indent=1))
    elif args.only == "cfg4":
        print(json.dumps(run_cfg4(ctx, torch, print, steps=args.steps, cpu=not args.no_cpu),
                         indent=1))
    else:
        print(json.dumps(run(ctx, torch, print), indent=1))
