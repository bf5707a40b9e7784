#!/usr/bin/env python3
"""Runs the witness-map transforms of a send-sized domain alone (no concurrent MSM streams), for rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blockmaze_amd import engine as e
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
a = np.random.default_rng(1).integers(0, 1 << 62, size=(m, 4), dtype=np.uint64); a[:, 3] >>= 2
for i in range(10):
    b = e.domain_transform(m, "cosetfft", a); c = e.domain_transform(m, "icosetfft", b)
print("ok" if np.array_equal(c, a) else "differs (expected with ZK_NTT_DEBUG)")
