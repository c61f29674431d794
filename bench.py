#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X RAW decompression core.

Metric (BASELINE.json): MPix/s decoded bit-exact + achieved HBM GB/s vs roofline.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: FRAMES frames of BASELINE configs[1]
(UncompressedDecompressor, 14-bit packed MSB, 8192x5464) decoded by ONE plan
launch through the C-ABI (rawspeed_amd/librsx.so).  The batch is larger than
the 256 MiB Infinity Cache on purpose, so the rate is an HBM rate.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

N > 1: one process per GPU.  Started under a launcher (torchrun: WORLD_SIZE set) the
process is one rank; started bare, `--gpus N` re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1.  Every rank decodes
its own shard of independent frames -- weak scaling for the headline leg, no data-path
collective; RCCL carries the barrier / max-reduction of the timing and, for the
batched-LJPEG leg (BASELINE configs[4]: 256 frames sharded with dist.shard_range over
the ranks), optionally the distribution of the packed batch from rank 0
(with N > 1 all three ways in one run -- every rank's own shard, grouped send/recv of
each rank's shard from rank 0, one RCCL broadcast of the whole batch -- each timed apart
from the decode and reported as `input_distribution`; --scatter / --broadcast: that one only).

Rank 0 prints ONE JSON line (kept under 6 KB: the driver holds a tail of stdout).  At
every N it carries
  roofline      -- dominant kernel: algorithmic bytes / hipEvent-measured launch time of
                   rank 0's launches (per GPU), plus the measured copy ceiling of the device
                   for the same bytes
  cpu_baseline  -- the unmodified reference (oracle/_ref) timed on this host's cores (rank 0)
  ljpeg.cfg5_batch_8192x5464 -- BASELINE configs[4] with its own in-run `roofline`
and at N=1 also
  ljpeg         -- summary of the LJPEG configs (cfg 3 / cfg 4 / cfg 5, clipped highlights,
                   uniform-random data): ms, GPix/s, fraction of the HBM peak, per-kernel
                   ms, CPU baselines, and -- replayed from profiles/ and labelled so -- the
                   VALU issue fraction and the measured HBM traffic of the pipeline
Everything else measured the same way (bench_ljpeg.py: configs[0], single-frame latency of
configs[1], the fixed-layout unpack entry points, Canon sRaw + Cr2sRawInterpolator, Nikon,
Hasselblad, Sony ARW1, the host-pointer path, the full LJPEG legs) goes to
bench_extra.json next to this file (and to stderr); the line names it in `extra_file`.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable

CFG1 = dict(w=4096, h=3072, bps=12, order=0)  # BASELINE configs[0]: BitOrder::LSB
CFG2 = dict(w=8192, h=5464, bps=14, order=1)  # BASELINE configs[1]: BitOrder::MSB
CFG5_TOTAL_FRAMES = 256                       # BASELINE configs[4]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-cfg5", action="store_true",
                    help="skip the batched-LJPEG-frames leg (BASELINE configs[4])")
    ap.add_argument("--cfg5-total-frames", type=int, default=CFG5_TOTAL_FRAMES,
                    help="frames of the whole LJPEG batch, sharded over the ranks")
    ap.add_argument("--cfg5-distinct", type=int, default=32,
                    help="different frames synthesised for the LJPEG batch")
    ap.add_argument("--broadcast", action="store_true",
                    help="cfg 5: rank 0 holds the packed batch and broadcasts it over RCCL")
    ap.add_argument("--scatter", action="store_true",
                    help="cfg 5: rank 0 holds the packed batch and sends every rank its shard")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: one rank per GPU under torchrun."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log("bench.py: no launcher in the environment, starting %d ranks: %s"
        % (args.gpus, " ".join(cmd[1:8])))
    return subprocess.call(cmd, env=env)


def out_pitch(w):
    return (w * 2 + 15) // 16 * 16  # RawImageData pitch (RawImage.cpp:80-83)


def make_frames(cfg, frames, seed0):
    """Packed strips of `frames` uniform-random frames (+ their pixels)."""
    from rawspeed_amd import synth
    w, h, bps, order = cfg["w"], cfg["h"], cfg["bps"], cfg["order"]
    packed, pxs = [], []
    for f in range(frames):
        px = synth.uniform(w * h, bps, seed0 + f).reshape(h, w)
        packed.append(synth.pack_rows(px, bps, order))
        pxs.append(px)
    return np.concatenate(packed), pxs


def unpack_jobs(cfg, frames):
    from rawspeed_amd import abi
    w, h, bps, order = cfg["w"], cfg["h"], cfg["bps"], cfg["order"]
    pitch, opitch = w * bps // 8, out_pitch(w)
    jobs = []
    for f in range(frames):
        j = abi.UnpackJob()
        j.desc = abi.UnpackDesc(0, 0, w, h, pitch, bps, order)
        j.in_offset, j.in_bytes = f * h * pitch, h * pitch
        j.img_offset = f * h * opitch
        j.img.pitch_bytes, j.img.dim_x, j.img.dim_y, j.img.cpp, j.img.is_cfa = \
            opitch, w, h, 1, 1
        jobs.append(j)
    return jobs


def frame_of(out_t, cfg, f):
    w, h = cfg["w"], cfg["h"]
    op = out_pitch(w)
    return out_t[f * h * op:(f + 1) * h * op].cpu().numpy().view(np.uint16) \
        .reshape(h, op // 2)[:, :w]


def profile_rounds():
    """profiles/rNN, newest first"""
    try:
        d = sorted((x for x in os.listdir(os.path.join(ROOT, "profiles"))
                    if x.startswith("r") and x[1:].isdigit()), key=lambda x: -int(x[1:]))
    except OSError:
        d = []
    return d


def pmc_traffic(frames):
    """HBM bytes per launch of the headline kernel from rocprofv3's PMC passes
    (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc runs of this same command).
    bench.py cannot run rocprofv3 on itself: the newest committed per-launch
    measurement under profiles/ is scaled to the batch size, and the JSON says so."""
    for rnd in profile_rounds():
        path = os.path.join(ROOT, "profiles", rnd, "unpack_pmc.json")
        try:
            with open(path) as f:
                d = json.load(f)
            return (int(d["traffic_bytes_per_launch"] * frames / 8),
                    "replayed from profiles/%s/unpack_pmc.json (rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE of this command, per launch, scaled to %d frames); not "
                    "measured in this run" % (rnd, frames))
        except Exception:
            continue
    return None, "no committed PMC measurement found"


def cpu_baseline_unpack(cfg, packed_frame, budget_s=20.0, expect=None):
    """The unmodified reference (oracle/_ref) on this host: 1 thread, then
    independent frames on all cores (the shape of rstest's omp-for over files).
    expect: the pixels the GPU produced for this frame -- the reference must agree."""
    from oracle_lib import Ref
    from rawspeed_amd import abi
    if not Ref.available():
        return None
    ref = Ref()
    w, h, bps, order = cfg["w"], cfg["h"], cfg["bps"], cfg["order"]
    d = abi.UnpackDesc(0, 0, w, h, w * bps // 8, bps, order)
    img = ref.image(w, h, 1)
    ref.unpack(d, packed_frame, img)  # warm-up / page touch
    agrees = None if expect is None else bool(np.array_equal(img.pixels(), expect))
    times = []
    t_end = time.perf_counter() + budget_s / 3
    while len(times) < 5 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        st = ref.unpack(d, packed_frame, img)
        times.append(time.perf_counter() - t0)
        assert st == 0
    single = w * h / min(times) / 1e6
    cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, ref.lib.ref_max_threads()))
    imgs = [ref.image(w, h, 1) for _ in range(nthreads)]
    ptrs = (C.c_void_p * nthreads)(*[i.h for i in imgs])
    a = np.ascontiguousarray(packed_frame)
    ins = (C.c_void_p * nthreads)(*[a.ctypes.data] * nthreads)
    ref.lib.ref_unpack_frames_parallel(nthreads, ptrs, C.byref(d), ins, a.size, nthreads)
    mt = []
    t_end = time.perf_counter() + budget_s * 2 / 3
    while len(mt) < 5 and (time.perf_counter() < t_end or len(mt) < 2):
        t0 = time.perf_counter()
        ref.lib.ref_unpack_frames_parallel(nthreads, ptrs, C.byref(d), ins, a.size,
                                           nthreads)
        mt.append(time.perf_counter() - t0)
    multi = nthreads * w * h / min(mt) / 1e6
    return {"value": round(multi, 1), "unit": "MPix/s", "cores": nthreads,
            "kind": "reference",
            "single_thread_value": round(single, 1),
            "reference_output_equals_gpu_output": agrees,
            "sample": "%d x one %dx%d %d-bit frame on %d threads (best of %d), "
                      "and 1 frame on 1 thread (best of %d); UncompressedDecompressor::"
                      "readUncompressedRaw of the unmodified reference (oracle/_ref, "
                      "clang -O3 -march=x86-64-v2)"
                      % (nthreads, w, h, bps, nthreads, len(mt), len(times))}


def time_plan(torch, grp, plan, inp, out, steps, warmup, stream):
    """max over ranks of the seconds for `steps` launches, bracketed by barriers"""
    for _ in range(warmup):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    grp.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    grp.barrier()
    return grp.max_over_ranks(time.perf_counter() - t0)


def small_unpack_leg(ctx, torch, cfg, frames, steps, stream, what, cpu):
    """one of the secondary uncompressed legs: bit-exact check of every frame + timing"""
    packed, pxs = make_frames(cfg, frames, 4242)
    w, h, bps = cfg["w"], cfg["h"], cfg["bps"]
    inp = torch.from_numpy(packed).cuda()
    out = torch.empty(frames * h * out_pitch(w), dtype=torch.uint8, device="cuda")
    plan = ctx.unpack_plan(unpack_jobs(cfg, frames))
    plan.run(inp.data_ptr(), out.data_ptr(), stream)
    rc, st, _ = plan.results()
    exact = rc == 0 and all(np.array_equal(frame_of(out, cfg, f), pxs[f])
                            for f in range(frames))
    for _ in range(10):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    alg = frames * (h * (w * bps // 8) + h * w * 2)
    res = {"workload": what, "frames_per_step": frames, "ms_per_step": round(dt * 1e3, 4),
           "mpix_per_s": round(frames * w * h / dt / 1e6, 1), "bit_exact": bool(exact),
           "algorithmic_bytes_per_step": alg,
           "achieved_gbps": round(alg / dt / 1e9, 1),
           "frac_of_hbm_peak": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4)}
    if cpu:
        try:
            res["cpu_baseline"] = cpu_baseline_unpack(cfg, packed[:h * (w * bps // 8)],
                                                      budget_s=8.0, expect=pxs[0])
        except Exception as e:
            res["cpu_baseline"] = {"error": repr(e)}
    return res


def cfg5_leg(args, ctx, torch, grp, rank, n_gpus, stream):
    """BASELINE configs[4]: a batch of 256 independent 8192x5464 LJPEG frames sharded over
    the ranks with dist.shard_range(256, N, rank) -- 32 per GPU on the 8 GPUs of a node,
    all 256 on one (the batch is fixed: strong scaling for this leg).  The packed batch
    is either synthesised by every rank for itself or, with --broadcast / --scatter,
    handed out by rank 0 over RCCL, timed separately from the decode."""
    import bench_ljpeg
    from rawspeed_amd import dist as rdist
    total = args.cfg5_total_frames
    lo, hi = rdist.shard_range(total, n_gpus, rank)
    f5 = hi - lo
    # global frame g of the batch is distinct frame (g + g // D) % D (seed 1000 + that):
    # D = 32 different frames (seeds 1000 + frame), and the shards of the
    # ranks differ from one another
    plan5, inp5, out5, meta = bench_ljpeg.make_cfg5_plan(ctx, torch, f5, seed0=1000,
                                                         distinct=args.cfg5_distinct,
                                                         first_frame=lo)
    # kernel-only scaling and distribution-inclusive scaling, reported apart.
    # With N > 1 and no flag the packed batch reaches the ranks all three ways in this one
    # run (rawspeed_amd.dist.distribute_all_modes): every rank's own shard, rank 0 sending
    # each rank ITS shard, rank 0 broadcasting the whole batch -- each must deliver the very
    # bytes the rank's plan expects, and the shard each mode delivered is decoded and
    # checked below.  --scatter / --broadcast restrict the run to that one exchange.
    dist_info = {"own_shard": {"what": "every rank synthesises its own shard: no exchange",
                               "ms": 0.0, "bytes": 0}}
    delivered = {}
    if grp.enabled:
        modes = (("own_shard", "broadcast") if args.broadcast else
                 ("own_shard", "scatter") if args.scatter else rdist.MODES)
        dist_info, delivered = rdist.distribute_all_modes(
            grp, total, lambda g: bench_ljpeg.cfg5_frame_bytes(meta, g),
            lambda units: bench_ljpeg.cfg5_assemble(torch, meta, units),
            lambda n: torch.empty(n, dtype=torch.uint8, device="cuda"), inp5,
            lambda a, b: a.numel() == b.numel() and bool(torch.equal(a, b)),
            sync=torch.cuda.synchronize, modes=modes)
        delivered.pop("own_shard", None)
    plan5.run(inp5.data_ptr(), out5.data_ptr(), stream)
    rc5, st5, cons5 = plan5.results()
    ref_frames = cpu5 = None
    if rank == 0 and not args.no_cpu_baseline:
        k = meta["distinct"]
        ref_frames, cpu5 = bench_ljpeg.ref_scan_baseline(
            0, [b[0] for b in meta["blobs"]], [b[1] for b in meta["blobs"]],
            meta["W"], meta["H"], "LJpegDecompressor::decode")
    exact5 = rc5 == 0 and bench_ljpeg.check_cfg5(out5, meta, cons5, f5, ref_frames)
    # ... and the shard as each exchange delivered it: decoded from THAT buffer, every frame checked
    for mode, shard in delivered.items():
        out5.zero_()
        plan5.run(shard.data_ptr(), out5.data_ptr(), stream)
        rcm, _, consm = plan5.results()
        okm = rcm == 0 and bench_ljpeg.check_cfg5(out5, meta, consm, f5, None)
        dist_info[mode]["decoded_bit_exact"] = bool(
            grp.sum_over_ranks(1.0 if okm else 0.0) == n_gpus)
        exact5 = exact5 and okm
    delivered.clear()
    k5 = 5
    plan5.set_timing(True)   # (event pool made outside the timed steps)
    plan5.set_timing(False)
    for _ in range(2):
        plan5.run(inp5.data_ptr(), out5.data_ptr(), stream)
    plan5.set_timing(True)
    dt5 = time_plan(torch, grp, plan5, inp5, out5, k5, 0, stream) / k5
    kt5 = ktab5 = None
    try:
        tab = plan5.kernel_table()   # (before kernel_time(), which resets the totals)
        ktab5 = {n: round(ms, 4) for n, ms in tab[0]} if tab else None
        kt5 = plan5.kernel_time()
    except Exception as e:
        log("cfg5 kernel table: %r" % (e,))
    plan5.set_timing(False)
    all_exact = grp.sum_over_ranks(1.0 if exact5 else 0.0) == n_gpus
    W5, H5 = meta["W"], meta["H"]
    for mode, rec in dist_info.items():
        # the batch's rate with that exchange in front of the decode (kernel-only: own_shard)
        rec["mpix_per_s_incl_distribution"] = round(
            total * W5 * H5 / (dt5 + rec.get("ms", 0.0) * 1e-3) / 1e6, 1)
    res = {
        "workload": "%d independent 8192x5464 LJPEG frames (2 components, predictor 1; "
                    "BASELINE configs[4]) sharded with shard_range over %d GPU(s): %d on "
                    "this rank, one plan launch per step" % (total, n_gpus, f5),
        "scaling": "strong (the batch is %d frames at every N)" % total,
        "mpix_per_s": round(total * W5 * H5 / dt5 / 1e6, 1),
        "ms_per_step": round(dt5 * 1e3, 3),
        "bit_exact": bool(all_exact),
        "bit_exact_against": "every frame of every rank against its source image"
                             + (" and oracle/_ref" if ref_frames is not None else ""),
        "entropy_bits_per_px": round(meta["bits_per_px"], 3),
        "frames_on_this_rank": f5,
        "distinct_frames": meta["distinct"],
        "achieved_gbps_whole_pipeline_per_gpu": round(meta["alg_bytes"] / dt5 / 1e9, 1),
        "frac_of_hbm_peak": round(meta["alg_bytes"] / dt5 / 1e9 / HBM_PEAK_GBPS, 4),
        "input_distribution": dist_info,
    }
    # the rank's own shard against the roofline, measured in this run (rank 0's events)
    if ktab5:
        res["kernels_ms"] = ktab5
    if kt5:
        res["dominant_kernel"] = {"name": kt5[0], "avg_ms": round(kt5[1], 4)}
    bench_ljpeg._roofline(res, meta["alg_bytes"], dt5, kt5)
    res["roofline"]["frames_on_this_rank"] = f5
    if cpu5:
        res["cpu_baseline"] = cpu5
    return res


def replayed_ljpeg_counters():
    """VALU issue fraction and HBM traffic of the cfg-3 LJPEG pipeline from the newest
    committed rocprofv3 PMC passes (scripts/pmc_ljpeg.sh, pmc_ljpeg_traffic.sh); bench.py
    cannot run rocprofv3 on itself."""
    out = {}
    for rnd in profile_rounds():
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "ljpeg_traffic", "ljpeg_traffic.json")) as f:
                t = json.load(f)
            out["traffic_over_algorithmic"] = t["traffic_over_algorithmic"]
            out["traffic_source"] = "replayed from profiles/%s/ljpeg_traffic/ljpeg_traffic.json " \
                "(rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, cfg 3, 8 frames); not measured in this run" % rnd
            break
        except Exception:
            continue
    for rnd in profile_rounds():
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "ljpeg_pmc", "ljpeg_pmc.json")) as f:
                t = json.load(f)
            # (a fraction of the issue slots cannot exceed 1: round 5's file had the PROBE
            # instantiation's time under the main kernel's counters and said 1.1-2.0; such a
            # file is not replayed, and the line says why)
            bad = {k: v for k, v in t["valu_issue_frac"].items()
                   if not all(0.0 <= float(x) <= 1.0 for x in (v if isinstance(v, list) else [v]))}
            if bad:
                out["valu_issue_frac_rejected"] = "profiles/%s/ljpeg_pmc/ljpeg_pmc.json holds a " \
                    "fraction > 1 (%s): counters and kernel time of different instantiations" % (
                        rnd, ", ".join(sorted(bad)))
                continue
            out["valu_issue_frac"] = t["valu_issue_frac"]
            out["valu_issue_source"] = "replayed from profiles/%s/ljpeg_pmc/ljpeg_pmc.json: %s" % (
                rnd, t.get("how", ""))
            break
        except Exception:
            continue
    # what bounds the single-pass kernel: workgroups x lifetime / resident slots, and how
    # often a symbol is parsed (profiles/rNN/ljpeg_limiter.json, scripts/ljpeg_limiter.py:
    # phase stamps of an experiment build + the PMC instruction counts + the ISA's resources)
    for rnd in profile_rounds():
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "ljpeg_limiter.json")) as f:
                t = json.load(f)
            for k in ("parses_per_symbol", "lane_instr_per_symbol", "wg_lifetime_us",
                      "resident_wg_per_cu", "wg_phases_us"):
                if k in t:
                    out[k] = t[k]
            out["limiter_source"] = "replayed from profiles/%s/ljpeg_limiter.json (%s); not " \
                "measured in this run" % (rnd, t.get("how", ""))
            break
        except Exception:
            continue
    return out


def ljpeg_summary(extra):
    """the LJPEG legs and the SURVEY 8(f) legs of `extra`, compressed for the one-line JSON.
    Every entry carries a `roofline` measured IN THIS RUN -- alg_MB: algorithmic bytes of a
    step, frac: of the 8 TB/s HBM peak over the whole step (wall clock of the timed steps),
    kernel / kernel_ms: the dominant kernel and its hipEvent average --; what comes from
    committed rocprofv3 PMC passes sits apart under `replayed`.  (Full objects: bench_extra.json.)"""
    def roof(d):
        r = d.get("roofline")
        if not isinstance(r, dict):
            return None
        k = r.get("kernel")
        return {"alg_MB": round(r.get("algorithmic_bytes", 0) / 1e6, 1), "frac": r.get("frac"),
                "kernel": k.replace("lj_", "").replace("_kernel", "")
                           .replace("legacy reconstruction (K5 + K6)", "legacy_recon") if k else None,
                "kernel_ms": round(r["avg_kernel_ms"], 3) if r.get("avg_kernel_ms") else None}

    def leg(d, cpu=True, kernels=True):
        if not isinstance(d, dict) or "ms_per_step" not in d:
            return d if isinstance(d, dict) and "error" in d else None
        r = {"ms": d["ms_per_step"], "gpix_per_s": round(d["mpix_per_s"] / 1e3, 1),
             "bit_exact": d.get("bit_exact"), "roofline": roof(d)}
        if kernels and "kernels_ms" in d:
            r["kernels_ms"] = {k.replace("lj_", "").replace("_kernel", ""): round(v, 3)
                               for k, v in d["kernels_ms"].items()}
        c = d.get("cpu_baseline")
        if cpu and isinstance(c, dict) and "value" in c:
            r["cpu_ref_mpix"] = {"threads_%d" % c["cores"]: c["value"],
                                 "threads_1": c.get("single_thread_value")}
            if "value_threads4" in c:
                r["cpu_ref_mpix"]["threads_4"] = c["value_threads4"]
        return r
    s = {
        "cfg3_cr2_6720x4480_8frames": leg(extra.get("cfg3_cr2_6720x4480")),
        "cfg4_dng_2x2_tiles_8192x5464_1frame": leg(extra.get("cfg4_dng_tiles_8192x5464")),
        "cfg5_batch_8192x5464": leg(extra.get("cfg5_ljpeg_frames_batch")),
        "cfg3_clipped_highlights": leg(extra.get("cfg3_clipped_highlights"), cpu=False, kernels=False),
        "cfg3_uniform_random_14bit": leg(extra.get("cfg3_uniform_random_14bit"), cpu=False, kernels=False),
        "ljpeg_3comp_8192x5464": leg(extra.get("ljpeg_3comp_8192x5464"), cpu=False, kernels=False),
        "ljpeg_3comp_3tables_8192x5464": leg(extra.get("ljpeg_3comp_3tables_8192x5464"), cpu=False, kernels=False),
        "ljpeg_4comp_1table_8192x5464": leg(extra.get("ljpeg_4comp_1table_8192x5464"), cpu=False, kernels=False),
        "ljpeg_4comp_4tables_8192x5464": leg(extra.get("ljpeg_4comp_4tables_8192x5464"), cpu=False, kernels=False),
    }
    c4 = extra.get("cfg4_dng_tiles_8192x5464")
    if isinstance(c4, dict):
        for k in ("two_tables", "two_tables_256x256_tiles", "overhang_8189x5462",
                  "restart_intervals"):
            if isinstance(c4.get(k), dict) and "ms_per_step" in c4[k]:
                s["cfg4_" + k] = leg(c4[k], cpu=False, kernels=False)
    c5 = extra.get("cfg5_ljpeg_frames_batch")
    if isinstance(c5, dict) and s.get("cfg5_batch_8192x5464"):
        s["cfg5_batch_8192x5464"]["frames_on_this_rank"] = c5.get("frames_on_this_rank")
        s["cfg5_batch_8192x5464"]["distinct_frames"] = c5.get("distinct_frames")
        s["cfg5_batch_8192x5464"]["input_distribution"] = c5.get("input_distribution")
    # the other decompressors, one line each
    f = {}
    nk = extra.get("nikon_lossless14_6016x4016")
    if isinstance(nk, dict) and isinstance(nk.get("curve_dither"), dict):
        f["nikon"] = leg(nk["curve_dither"], cpu=False, kernels=False)
    for key, name in (("hasselblad_8272x6200", "hasselblad"), ("sony_arw1_3881x2608", "sony_arw1"),
                      ("pentax_7392x4950", "pentax"), ("samsung_v1_5472x3648", "samsung_v1"),
                      ("samsung_v2_6480x4320", "samsung_v2"), ("cr2_sraw1_3960x2640", "cr2_sraw_decode")):
        f[name] = leg(extra.get(key), cpu=False, kernels=False)
    sr = extra.get("cr2_sraw1_3960x2640")
    if isinstance(sr, dict) and isinstance(sr.get("interpolate"), dict):
        f["cr2_sraw_interpolate"] = leg(sr["interpolate"], cpu=False, kernels=False)
    s["survey_8f"] = {k: v for k, v in f.items() if v}
    # (replayed figures: the numbers and the FILE they come from; phases, parse counts and the
    # long provenance texts stay in that file and in bench_extra.json -- the line is a tail
    # of stdout for the driver, 6 KB at most)
    rep = replayed_ljpeg_counters()
    short = {k: rep[k] for k in ("traffic_over_algorithmic", "valu_issue_frac",
                                 "valu_issue_frac_rejected",
                                 "lane_instr_per_symbol", "wg_lifetime_us") if k in rep}
    import re as _re
    srcs = sorted({m for v in rep.values() if isinstance(v, str)
                   for m in _re.findall(r"profiles/r\d+/[\w/.]+", v)})
    short["source"] = "committed rocprofv3 passes, not measured in this run: " + ", ".join(srcs)
    s["replayed"] = short
    return s


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    from rawspeed_amd import dist as rdist
    world, rank, local_rank = rdist.env_world()
    torch.cuda.set_device(local_rank)
    grp = rdist.Group(backend="nccl", device=torch.device("cuda", local_rank))
    n_gpus = world if grp.enabled else 1
    if args.gpus != n_gpus and rank == 0:
        log("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, world, n_gpus))

    import __graft_entry__ as ge
    ge.build()
    from rawspeed_amd import capi

    ctx = capi.Context(local_rank)
    F = args.frames
    w, h, bps = CFG2["w"], CFG2["h"], CFG2["bps"]
    opitch = out_pitch(w)
    packed, pxs = make_frames(CFG2, F, 1000 + 100 * rank)
    inp = torch.from_numpy(packed).cuda()
    out = torch.empty(F * h * opitch, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    plan = ctx.unpack_plan(unpack_jobs(CFG2, F))

    # untimed: first touch of the buffers, bit-exactness of the path being timed --
    # every frame against the pixels it was packed from (and, at N=1, the reference)
    plan.run(inp.data_ptr(), out.data_ptr(), stream)
    rc, st, _ = plan.results()
    assert rc == 0, (rc, st)
    bit_exact = all(np.array_equal(frame_of(out, CFG2, f), pxs[f]) for f in range(F))
    assert bit_exact, "GPU output differs from the packed source"
    plan.set_timing(True)  # pre-creates the event pool (slow on ROCm) outside the timed region
    plan.set_timing(False)
    for _ in range(args.warmup):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    plan.set_timing(True)
    grp.barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    grp.barrier()
    my_elapsed = time.perf_counter() - t_start
    ktime = plan.kernel_time()
    plan.set_timing(False)
    elapsed = grp.max_over_ranks(my_elapsed)
    per_rank = grp.gather_objects(round(F * w * h * args.steps / my_elapsed / 1e6, 1), dst=0)
    # copy ceiling for the same read:write mix (outside the timed region): a plain
    # streaming kernel over the very same buffers
    copy_ms = None
    if rank == 0:
        try:
            copy_ms = ctx.probe_stream_copy(inp.data_ptr(), F * h * (w * bps // 8),
                                            out.data_ptr(), F * h * opitch, stream, reps=20)
            plan.run(inp.data_ptr(), out.data_ptr(), stream)  # the probe overwrote `out`
        except Exception as e:
            log("copy probe failed: %r" % (e,))

    pix_per_step = F * w * h * n_gpus
    value = pix_per_step * args.steps / elapsed / 1e6
    result = {
        "metric": "MPix/s decoded (bit-exact)",
        "value": round(value, 1),
        "unit": "MPix/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "bit_exact": bool(bit_exact),
        "rccl_ranks": world if grp.enabled else 0,
        "per_rank_mpix_per_s": per_rank,
        "config": {
            "workload": "UncompressedDecompressor 14-bit packed MSB 8192x5464 "
                        "(BASELINE configs[1]), %d independent frames per GPU per step, "
                        "inputs and outputs resident in HBM" % F,
            "frames_per_gpu": F,
            "parallelism": "frames sharded across %d GPU(s), one process per GPU, no "
                           "data-path collective (RCCL: barrier + max-reduction of the "
                           "timing)" % n_gpus,
        },
    }

    # BASELINE configs[4]: batch of independent LJPEG frames sharded over the GPUs
    cfg5 = None
    if not args.no_cfg5:
        try:
            del out, inp
            torch.cuda.empty_cache()
            cfg5 = cfg5_leg(args, ctx, torch, grp, rank, n_gpus, stream)
        except Exception as e:  # the headline number must survive
            cfg5 = {"error": repr(e)}

    # rank 0 at every N: `roofline` from ITS in-run kernel time (per GPU: every rank runs the
    # same launch over its own frames) and `cpu_baseline`; the secondary legs at N = 1 only
    if rank == 0:
        alg_bytes = F * (h * (w * bps // 8) + h * w * 2)  # packed read once + u16 written once
        if ktime:
            name, avg_ms, n = ktime
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic, source = pmc_traffic(F)
            result["roofline"] = {
                "bound": "hbm", "kernel": name,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "traffic_source": source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": round(avg_ms, 5), "launches_timed": n,
                "scope": "one GPU (rank 0's launches); the job's rate is `value`",
            }
            if copy_ms:
                ceil = alg_bytes / (copy_ms * 1e-3) / 1e9
                result["roofline"]["copy_ceiling"] = {
                    "gbps": round(ceil, 1), "avg_kernel_ms": round(copy_ms, 5),
                    "frac_of_ceiling": round(achieved / ceil, 4),
                    "what": "plain 16-byte load / non-temporal store kernel over the same "
                            "buffers (same bytes in, same bytes out)",
                }
        if not args.no_extra and n_gpus == 1:
            extra = {}
            try:
                extra["cfg1_12bit_lsb_4096x3072"] = small_unpack_leg(
                    ctx, torch, CFG1, 8, 50, stream,
                    "UncompressedDecompressor 12-bit packed LSB 4096x3072 (BASELINE "
                    "configs[0]), 8 frames per step", not args.no_cpu_baseline)
                extra["cfg2_single_frame_latency"] = small_unpack_leg(
                    ctx, torch, CFG2, 1, 200, stream,
                    "one 8192x5464 14-bit MSB frame per launch (latency of a single "
                    "decode; 168 MB fit the Infinity Cache)", False)
            except Exception as e:
                extra["unpack_legs_error"] = repr(e)
            try:
                import bench_ljpeg
                extra.update(bench_ljpeg.run(ctx, torch, log))
            except Exception as e:  # the headline number must survive
                extra["error"] = repr(e)
            result["extra"] = extra
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline_unpack(
                    CFG2, packed[:h * (w * bps // 8)], expect=pxs[0])
            except Exception as e:
                result["cpu_baseline"] = {"error": repr(e)}
    if cfg5 is not None:
        result.setdefault("extra", {})["cfg5_ljpeg_frames_batch"] = cfg5
    if rank == 0:
        extra = result.pop("extra", None)
        if extra is not None:
            result["ljpeg"] = ljpeg_summary(extra)
            path = os.path.join(ROOT, "bench_extra.json")
            try:
                with open(path, "w") as f:
                    json.dump(extra, f, indent=1)
                result["extra_file"] = "bench_extra.json (every other leg, full detail)"
            except OSError as e:
                result["extra_file"] = "not written: %r" % (e,)
            log("bench_extra: " + json.dumps(extra))
        print(json.dumps(result), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
