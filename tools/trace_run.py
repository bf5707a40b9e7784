#!/usr/bin/env python3
"""A dozen resident proofs of one circuit, for a kernel trace:   rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/trace_run.py [mint|redeem|send|deposit] [depth]
then `python tools/timeline.py out/.../*kernel_trace.csv 3` for the per-stream timeline of one proof."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ["ZK_TRACE_TIMES"] = "1"
import numpy as np
from blockmaze_amd import engine as e
import workload as w
kind = sys.argv[1] if len(sys.argv) > 1 else "send"; depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hx = lambda args: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in args]
tmp = tempfile.mkdtemp(); pk, vk, wp = tmp + "/pk.txt", tmp + "/vk.txt", tmp + "/w.bin"; e.keygen(kind, pk, vk, seed=7, tree_depth=depth)
if kind == "send": e.witness_send(*hx(w.send_args(w.send_instance(1))), wp)
elif kind in ("mint", "redeem"): e.witness_mint_redeem(kind == "redeem", *hx(w.mint_args(w.mint_instance(1, redeem=(kind == "redeem")))), wp)
else: d = w.deposit_instance(1); e.witness_deposit(*hx(w.deposit_args(d)), "".join("0x" + l.hex() for l in d["leaves"]), len(d["leaves"]), "0x" + d["sk"].hex(), wp, tree_depth=depth)
b = open(wp, "rb").read(); k = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); z = np.frombuffer(b, dtype=np.uint64, count=4 * k, offset=8).reshape(k, 4).copy()
p = e.Prover(pk); p.set_witness(z)
for _ in range(12): p.prove_resident()
p.close()
