#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X RAW decompression core.

Metric (BASELINE.json): MPix/s decoded bit-exact + achieved HBM GB/s vs roofline.

A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM: FRAMES frames of BASELINE configs[1]
(UncompressedDecompressor, 14-bit packed MSB, 8192x5464) decoded by ONE plan
launch through the C-ABI (rawspeed_amd/librsx.so).  The batch is larger than
the 256 MiB Infinity Cache on purpose, so the rate is an HBM rate.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

N>1: one process per GPU (torchrun), every rank decodes its own shard of
independent frames (weak scaling, no data-path collective; RCCL is used for the
barrier / max-reduction of the timing and, with --broadcast, to distribute the
packed buffer from rank 0 over xGMI, timed separately).

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline      -- dominant kernel: algorithmic bytes / hipEvent-measured launch time,
                   plus the measured copy ceiling of the device for the same bytes
  cpu_baseline  -- the unmodified reference (oracle/_ref) timed on this host's cores
  extra         -- the other legs measured the same way (bench_ljpeg.py): the LJPEG
                   configs (cfg 3 / cfg 4 / cfg 5), the fixed-layout unpack entry
                   points, Canon sRaw + Cr2sRawInterpolator, Nikon, Hasselblad, Sony ARW1
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable

CFG2 = dict(w=8192, h=5464, bps=14, order=1)  # BitOrder::MSB


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
    ap.add_argument("--cfg5-frames", type=int, default=32, help="LJPEG frames per GPU")
    ap.add_argument("--broadcast", action="store_true",
                    help="N>1: rank 0 synthesises the packed batch and broadcasts it over RCCL")
    return ap.parse_args()


def out_pitch():
    return (CFG2["w"] * 2 + 15) // 16 * 16  # RawImageData pitch (RawImage.cpp:80-83)


def make_frames(frames, seed0):
    """Packed strips of `frames` uniform-random 14-bit frames (+ frame 0's pixels)."""
    from rawspeed_amd import synth
    w, h, bps, order = CFG2["w"], CFG2["h"], CFG2["bps"], CFG2["order"]
    packed, px0 = [], None
    for f in range(frames):
        px = synth.uniform(w * h, bps, seed0 + f).reshape(h, w)
        packed.append(synth.pack_rows(px, bps, order))
        if f == 0:
            px0 = px
    return np.concatenate(packed), px0


def unpack_jobs(frames):
    from rawspeed_amd import abi
    w, h, bps, order = CFG2["w"], CFG2["h"], CFG2["bps"], CFG2["order"]
    pitch, opitch = w * bps // 8, out_pitch()
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


def pmc_traffic(frames):
    """HBM bytes per launch from the rocprofv3 PMC passes of this kernel
    (profiles/r01/unpack_pmc.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    --pmc runs of this same command).  bench.py cannot run rocprofv3 on itself,
    so the committed per-launch measurement is scaled to the batch size."""
    path = os.path.join(ROOT, "profiles", "r01", "unpack_pmc.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return int(d["traffic_bytes_per_launch"] * frames / 8)
    except Exception:
        return None


def cpu_baseline_unpack(packed_frame, budget_s=20.0):
    """The unmodified reference (oracle/_ref) on this host: 1 thread, then
    independent frames on all cores (the shape of rstest's omp-for over files)."""
    from oracle_lib import Ref
    from rawspeed_amd import abi
    if not Ref.available():
        return None
    ref = Ref()
    w, h, bps, order = CFG2["w"], CFG2["h"], CFG2["bps"], CFG2["order"]
    d = abi.UnpackDesc(0, 0, w, h, w * bps // 8, bps, order)
    img = ref.image(w, h, 1)
    ref.unpack(d, packed_frame, img)  # warm-up / page touch
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
            "sample": "%d x one 8192x5464 14-bit MSB frame on %d threads (best of %d), "
                      "and 1 frame on 1 thread (best of %d); UncompressedDecompressor::"
                      "readUncompressedRaw of the unmodified reference (oracle/_ref, "
                      "clang -O3 -march=x86-64-v2)" % (nthreads, nthreads, len(mt), len(times))}


def main():
    args = parse()
    import torch
    from rawspeed_amd import dist as rdist
    world, rank, local_rank = rdist.env_world()
    torch.cuda.set_device(local_rank)
    grp = rdist.Group(backend="nccl", device=torch.device("cuda", local_rank))
    distributed = grp.enabled
    dist = grp.dist if distributed else None
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        log("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, world, n_gpus))

    import __graft_entry__ as ge
    ge.build()
    from rawspeed_amd import capi

    ctx = capi.Context(local_rank)
    F = args.frames
    w, h, bps = CFG2["w"], CFG2["h"], CFG2["bps"]
    opitch = out_pitch()
    jobs = unpack_jobs(F)
    bcast_ms = None
    if distributed and args.broadcast:
        # rank 0 synthesises the batch; everyone receives it over RCCL / xGMI
        if rank == 0:
            packed, px0 = make_frames(F, 1000)
            inp = torch.from_numpy(packed).cuda()
        else:
            packed, px0 = None, None
            inp = torch.empty(F * h * (w * bps // 8), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        dist.broadcast(inp, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb) * 1e3
    else:
        packed, px0 = make_frames(F, 1000 + 100 * rank)
        inp = torch.from_numpy(packed).cuda()
    out = torch.empty(F * h * opitch, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    plan = ctx.unpack_plan(jobs)

    barrier = grp.barrier

    # untimed: first-touch of the buffers, bit-exactness of the path being timed
    plan.run(inp.data_ptr(), out.data_ptr(), stream)
    rc, st, _ = plan.results()
    assert rc == 0, (rc, st)
    bit_exact = None
    if px0 is not None:
        got = out[:h * opitch].cpu().numpy().view(np.uint16).reshape(h, opitch // 2)[:, :w]
        bit_exact = bool(np.array_equal(got, px0))
        assert bit_exact, "GPU output differs from the packed source"
    plan.set_timing(True)  # pre-creates the event pool (slow on ROCm) outside the timed region
    plan.set_timing(False)
    for _ in range(args.warmup):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    plan.set_timing(True)
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        plan.run(inp.data_ptr(), out.data_ptr(), stream)
    barrier()
    elapsed = time.perf_counter() - t_start
    ktime = plan.kernel_time()
    plan.set_timing(False)
    elapsed = grp.max_over_ranks(elapsed)
    # copy ceiling for the same read:write mix (outside the timed region): a plain
    # streaming kernel over the very same buffers
    copy_ms = None
    if rank == 0 and n_gpus == 1:
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
        "bit_exact": bit_exact,
        "config": {
            "workload": "UncompressedDecompressor 14-bit packed MSB 8192x5464 "
                        "(BASELINE configs[1]), %d independent frames per GPU per step, "
                        "inputs and outputs resident in HBM" % F,
            "frames_per_gpu": F,
            "parallelism": "frames sharded across %d GPU(s), no data-path collective"
                           % n_gpus,
        },
    }
    if bcast_ms is not None:
        result["config"]["input_broadcast_ms"] = round(bcast_ms, 2)

    # BASELINE configs[4]: batch of independent LJPEG frames sharded over the GPUs
    # (256 frames on 8 GPUs = 32 per GPU; weak scaling, same per-GPU shard at any N)
    cfg5 = None
    if not args.no_cfg5:
        try:
            import bench_ljpeg
            del out, inp
            torch.cuda.empty_cache()
            f5 = args.cfg5_frames
            plan5, inp5, out5, meta = bench_ljpeg.make_cfg5_plan(ctx, torch, f5,
                                                                 seed0=1000 + 10 * rank)
            plan5.run(inp5.data_ptr(), out5.data_ptr(), stream)
            rc5, st5, cons5 = plan5.results()
            W5, H5 = meta["W"], meta["H"]
            op5 = bench_ljpeg.out_pitch(W5)
            got5 = out5[:op5 * H5].cpu().numpy().view(np.uint16).reshape(H5, op5 // 2)[:, :W5]
            exact5 = bool(rc5 == 0 and np.array_equal(got5, meta["src0"])
                          and cons5 == meta["lens"])
            for _ in range(2):
                plan5.run(inp5.data_ptr(), out5.data_ptr(), stream)
            barrier()
            t5 = time.perf_counter()
            k5 = 5
            for _ in range(k5):
                plan5.run(inp5.data_ptr(), out5.data_ptr(), stream)
            barrier()
            dt5 = grp.max_over_ranks((time.perf_counter() - t5) / k5)
            all_exact = grp.sum_over_ranks(1.0 if exact5 else 0.0) == n_gpus
            cfg5 = {
                "workload": "%d independent 8192x5464 LJPEG frames per GPU (2 components, "
                            "predictor 1), %d GPU(s), one plan launch per step" % (f5, n_gpus),
                "mpix_per_s": round(n_gpus * f5 * W5 * H5 / dt5 / 1e6, 1),
                "ms_per_step": round(dt5 * 1e3, 3),
                "bit_exact": bool(all_exact),
                "entropy_bits_per_px": round(meta["bits_per_px"], 3),
                "achieved_gbps_whole_pipeline_per_gpu": round(meta["alg_bytes"] / dt5 / 1e9, 1),
                "frac_of_hbm_peak": round(meta["alg_bytes"] / dt5 / 1e9 / HBM_PEAK_GBPS, 4),
            }
            del plan5, inp5, out5
        except Exception as e:  # the headline number must survive
            cfg5 = {"error": repr(e)}

    if rank == 0 and n_gpus == 1:
        alg_bytes = F * (h * (w * bps // 8) + h * w * 2)  # packed read once + u16 written once
        if ktime:
            name, avg_ms, n = ktime
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            result["roofline"] = {
                "bound": "hbm", "kernel": name,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": pmc_traffic(F),
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": round(avg_ms, 5), "launches_timed": n,
            }
            if copy_ms:
                ceil = alg_bytes / (copy_ms * 1e-3) / 1e9
                result["roofline"]["copy_ceiling"] = {
                    "gbps": round(ceil, 1), "avg_kernel_ms": round(copy_ms, 5),
                    "frac_of_ceiling": round(achieved / ceil, 4),
                    "what": "plain 16-byte load / non-temporal store kernel over the same "
                            "buffers (same bytes in, same bytes out)",
                }
        if not args.no_extra:
            try:
                import bench_ljpeg
                result["extra"] = bench_ljpeg.run(ctx, torch, log)
            except Exception as e:  # the headline number must survive
                result["extra"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline_unpack(
                    packed[:h * (w * bps // 8)])
            except Exception as e:
                result["cpu_baseline"] = {"error": repr(e)}
    if cfg5 is not None:
        result.setdefault("extra", {})["cfg5_ljpeg_frames_batch"] = cfg5
    if rank == 0:
        print(json.dumps(result), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
