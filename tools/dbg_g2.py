#!/usr/bin/env python3
"""debug: the G2 witness MSM (lanes on 29-bit limbs against the quad kernel) on growing sizes, against the oracle"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from blockmaze_amd import engine as e
    from oracle import pyoracle as o
    for n, consecutive in ((600, True), (5000, True), (20000, True), (20000, False), (136316, True)):
        g = o.SplitMix64(0xB2 + n); rng = np.random.default_rng(36 + n)
        if consecutive: P = o.g2_consecutive(g.field(), n)
        else:
            base = o.g2_consecutive(g.field(), 64); P = np.zeros((n, 16), dtype=np.uint64)
            for i in range(n): P[i] = base[(i * 37 + (i >> 6)) % 64]
            # distinct-ish multiples: scale block b by (b+1) through repeated use is not needed; repeats are fine for the general path, and rare per lane
        Z = np.zeros((n, 4), dtype=np.uint64); Z[:] = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 61) - 1)
        sel = rng.integers(0, 1000, size=n); Z[sel < 509] = 0; ones = (sel >= 509) & (sel < 967); Z[ones] = 0; Z[ones, 0] = 1; small = (sel >= 967) & (sel < 995); Z[small, 1:] = 0; Z[small, 0] &= np.uint64(0xffffffff)
        exp = o.msm_g2(P, Z, mixed=True); got = o.g2_from(e.msm(2, P, Z, 8, filter_ones=True))[0]
        only_ones = Z.copy(); only_ones[~ones] = 0; exp1 = o.msm_g2(P, only_ones, mixed=True); got1 = o.g2_from(e.msm(2, P, only_ones, 8, filter_ones=True))[0]
        no_ones = Z.copy(); no_ones[ones] = 0; exp2 = o.msm_g2(P, no_ones, mixed=True); got2 = o.g2_from(e.msm(2, P, no_ones, 8, filter_ones=True))[0]
        print("n %6d consecutive %d: all %s | ones only %s | no ones %s" % (n, consecutive, got == exp, got1 == exp1, got2 == exp2), "(got None)" if got is None else "", flush=True)
else:
    for v in ("1", "0"):
        print("== ZK_G2_LANES29=" + v, flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ZK_G2_LANES29=v), capture_output=True, text=True)
        print(r.stdout); print(r.stderr[-1500:])
