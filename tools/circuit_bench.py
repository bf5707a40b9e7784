#!/usr/bin/env python3
"""Per-circuit timings on one GPU: key generation, key load, proof (witness resident), for mint / redeem / send / deposit (depth 8) / deposit (depth 32)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from blockmaze_amd import engine as e
import workload as w
hx = lambda args: [("0x" + a.hex()) if isinstance(a, bytes) else a for a in args]
tmp = tempfile.mkdtemp(); wp = os.path.join(tmp, "w.bin")
def witness(kind, depth):
    if kind == "send": d = w.send_instance(1); e.witness_send(*hx(w.send_args(d)), wp)
    elif kind in ("mint", "redeem"): d = w.mint_instance(1, redeem=(kind == "redeem")); e.witness_mint_redeem(kind == "redeem", *hx(w.mint_args(d)), wp)
    else: d = w.deposit_instance(1); e.witness_deposit(*hx(w.deposit_args(d)), "".join("0x" + l.hex() for l in d["leaves"]), len(d["leaves"]), "0x" + d["sk"].hex(), wp, tree_depth=depth)
    b = open(wp, 'rb').read(); k = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); return np.frombuffer(b, dtype=np.uint64, count=4 * k, offset=8).reshape(k, 4).copy()
ALL = (("mint", 8), ("redeem", 8), ("send", 8), ("deposit", 8), ("deposit", 32))
only = set(sys.argv[1:])                                    # e.g. "deposit:32 send:8"; nothing = all five
for kind, depth in ALL:
    if only and ("%s:%d" % (kind, depth)) not in only: continue
    pk, vk = os.path.join(tmp, "pk.txt"), os.path.join(tmp, "vk.txt"); t0 = time.time(); e.keygen(kind, pk, vk, seed=7, tree_depth=depth); tk = time.time() - t0
    t0 = time.time(); p = e.Prover(pk); tl = time.time() - t0; p.close(); t0 = time.time(); p = e.Prover(pk); tc = time.time() - t0; z = witness(kind, depth); p.set_witness(z); p.prove_resident(); n = int(os.environ.get("ZK_CB_N", "20"))   # second load: from the container the first one left behind
    t0 = time.perf_counter(); dev = []
    for _ in range(n): proof = p.prove_resident(); dev.append(p.timings()["device_ms"])
    ms = 1e3 * (time.perf_counter() - t0) / n; t0 = time.perf_counter(); dev.sort(); dev_med = dev[len(dev) // 2]   # (the prover's own device clock: the figure that does not move with the host's mood)
    for _ in range(n): proof = p.prove(z)
    msh = 1e3 * (time.perf_counter() - t0) / n
    print("%-8s depth %2d: %8d variables, domain %8d | keygen %5.2f s | key load %5.2f s from text (%6.1f MB), %5.2f s from the container | %6.2f ms/proof resident = %6.1f /s (device clock, median: %.3f ms) | %6.2f ms/proof host buffer = %6.1f /s" % (kind, depth, p.n_vars, p.m, tk, tl, os.path.getsize(pk) / 1e6, tc, ms, 1e3 / ms, dev_med, msh, 1e3 / msh))
    if os.environ.get("ZK_CB_STAGES"):                       # per-stage HIP-event times of a few more proofs
        e.profile_enable(True)
        for _ in range(5): p.prove_resident()
        st = e.profile_report(); e.profile_enable(False)
        print("   stages (ms per proof): " + ", ".join("%s %.3f" % (k, v["ms_total"] / 5) for k, v in sorted(st.items())))
    p.close()
