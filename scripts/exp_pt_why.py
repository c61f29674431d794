"""Why does the single-pass kernel give a table-per-phase stream up?  (stats variant: RSX_LIB=.../librsx_stats.so RSX_DEBUG=1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cases as C
import gpu_util
import test_gpu_per_component_tables as T
from oracle_lib import Oracle
gpu, oracle = gpu_util.ctx(), Oracle()
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, cpp, index = T.PATTERNS[which] if which < 100 else (3, 3, [0, 0, 0])
rng = np.random.default_rng([808, n] + index)
prec = 14
tw, th = (1536 if cpp == 1 else 768), int(sys.argv[2]) if len(sys.argv) > 2 else 700
tables = T._tables(rng, max(index) + 1, prec)
px = T._sensor_like(rng, th, tw * cpp) if len(sys.argv) > 3 else C.smooth_image(rng, th, tw * cpp, prec, sigma=25.0)
d, data = T._stream(rng, px, n, prec, tables, index)
d.tile_x, d.tile_y, d.tile_w, d.tile_h = 0, 0, tw, th
try:
    plan, inp, out = T._decode(gpu, oracle, [(d, data)], tw, th, cpp)
    print("decode ok; kernels:", T._kernel_names(plan, inp, out))
except AssertionError as e:
    print("FAILED", str(e)[:300])
