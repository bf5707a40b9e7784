#!/usr/bin/env python3
"""Overflow fallback of the witness sort inside the prover: proofs of the golden fixtures with ZK_MSM_DIRECT_CAP=1 (every bucket region overflows, the general
MSM path takes over) must equal the reference prover's bytes, twice in a row on the same prover object.   ZK_MSM_DIRECT_CAP=1 python tools/dbg_overflow.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import numpy as np
d = os.path.join(ROOT, "tests", "golden") + "/groth16_small"; meta = json.load(open(d + "/meta.json")); z = o.load_witness(d + "/wit.bin"); p = e.Prover(d + "/pk.txt")
for i in range(3): print(i, p.prove(z, int(meta['r'], 16), int(meta['s'], 16)) == meta['proof'])
n = 500; g = o.SplitMix64(5); P = o.g1_consecutive(g.field(), n); rng = np.random.default_rng(3); Z = np.zeros((n, 4), dtype=np.uint64); sel = rng.integers(0, 10, size=n); Z[sel < 4, 0] = 1; big = sel >= 7; Z[big, 0] = rng.integers(2, 1 << 40, size=int(big.sum()), dtype=np.uint64)
print("msm", o.g1_from(e.msm(1, P, Z, 8, filter_ones=True))[0] == o.msm_g1(P, Z, mixed=True))
m = e.ResidentMsm(1, P, 8, True); m.set_scalars(Z); exp = o.msm_g1(P, Z, mixed=True)
for i in range(4): print("resident run", i, o.g1_from(m.run())[0] == exp)
