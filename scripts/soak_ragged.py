"""Soak of tests/test_gpu_host_path_ragged.py's six-thread case with a post-mortem: on any difference
from the oracle (status, consumed bytes, pixels) print what differs, then decode the same case three
more times on the calling thread and say whether it repeats.   RSX_FUZZ_BASE=<k> python scripts/soak_ragged.py"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_host_path_ragged as T
import gpu_util
from oracle_lib import HostImage, Oracle

gpu = gpu_util.ctx()
oracle = Oracle()
events = 0
for kind, lo, hi in (("ragged", 0, 90), ("middle", 0, 70), ("odd", 0, 50)):
    cases = [T.make_case(s, kind) for s in range(lo, hi)]
    wants = [T.oracle_image(oracle, c) for c in cases]
    results = [None] * len(cases)

    def job(i):
        def run():
            c = cases[i]
            img = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
            rc, st, cons = gpu.dng_decompress_ljpeg(c["descs"], c["datas"], img.view())
            results[i] = (img, rc, list(st), list(cons), threading.get_ident())
        return run
    T.run_threads([job(i) for i in range(len(cases))])
    for c, (want, so), (img, rc, st, cons, tid) in zip(cases, wants, results):
        ok_st = rc == 0 and st == [0] * len(st)
        ok_cons = cons == [s[1] for s in so]
        ok_px = np.array_equal(img.buf, want.buf)
        if ok_st and ok_cons and ok_px:
            continue
        events += 1
        print("EVENT %s seed %d base %d thread %x: rc %d st %s cons %s want %s pixels %s; in_bytes %s" % (
            kind, c["seed"], T.BASE, tid, rc, st, cons, [s[1] for s in so],
            "equal" if ok_px else T.describe(c, img, want), [d.size for d in c["datas"]]), flush=True)
        for k in range(3):
            img2 = HostImage(c["W"], c["H"], c["cpp"], is_cfa=c["cpp"] == 1)
            rc2, st2, cons2 = gpu.dng_decompress_ljpeg(c["descs"], c["datas"], img2.view())
            print("   again %d: rc %d st %s cons %s pixels %s" % (
                k, rc2, list(st2), list(cons2),
                "equal" if np.array_equal(img2.buf, want.buf) else T.describe(c, img2, want)), flush=True)
print("soak base %d: %d events" % (T.BASE, events))
