#!/usr/bin/env python3
"""Profiling driver: H-query-sized G1 MSM (2^18 - 1 points, full-width scalars) on resident bases.  python3 tools/msm_bench.py [iters] [window_bits]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from blockmaze_amd import engine as e
from oracle import pyoracle as o   # only to build test bases cheaply
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5; c = int(sys.argv[2]) if len(sys.argv) > 2 else 13
n = (1 << 18) - 1; P = o.g1_consecutive(12345, n); rng = np.random.default_rng(1); K = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); K[:, 3] >>= 2
m = e.ResidentMsm(1, P, c, False); m.set_scalars(K); m.run()
e.profile_enable(True); t0 = time.time()
for _ in range(iters): r = m.run()
dt = (time.time() - t0) / iters; rep = e.profile_report()
print("c=%d  %.3f ms per MSM" % (c, dt * 1e3), {k: round(v["ms_total"] / v["count"], 3) for k, v in rep.items()})
