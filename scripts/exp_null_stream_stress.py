"""Plan creation against a BUSY null stream (round 6).  hipMemset is asynchronous with respect to the
host on this runtime (scripts/repro/memset_async.hip); until round 6 a plan's creation zeroed six device
arrays with it and the plan's kernels, on a hipStreamNonBlocking stream, could run first.  Here a second
host thread keeps the null stream busy on purpose (torch's default stream IS the null stream:
torch.cuda._sleep; never more than BURST kernels of US microseconds queued), while LJPEG host calls of
small tiles are made one after the other and compared with the oracle.
   RSX_LIB=.../librsx_nosync.so BURST=4 US=10 python scripts/exp_null_stream_stress.py    (the old behaviour)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_host_path_ragged as T
import gpu_util
from oracle_lib import Oracle

gpu = gpu_util.ctx()
cases = T.small_cases(Oracle(), int(os.environ.get("CASES", "100")))
burst, us = int(os.environ.get("BURST", "4")), float(os.environ.get("US", "10"))
t0 = time.time()
calls, wrong = T.run_against_busy_null_stream(gpu, cases, int(os.environ.get("ROUNDS", "4")), burst, int(us * 2400))
for w in wrong[:4]:
    print("EVENT round %d seed %d: rc %d st %s cons %s want %d" % w)
print("null-stream stress (bursts of %d kernels of %g us): %d calls in %.1f s, %d wrong" % (burst, us, calls, time.time() - t0, len(wrong)))
