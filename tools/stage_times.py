#!/usr/bin/env python3
"""HIP-event stage clocks of the prover (zkgpu_profile_report) over N resident send proofs: python tools/stage_times.py [prefix] [N]   -> one line "stage ms ..." for the stages that start with prefix"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
prefix = sys.argv[1] if len(sys.argv) > 1 else ""; N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tmp = tempfile.mkdtemp(); pk, vk, wp = tmp + "/pk.txt", tmp + "/vk.txt", tmp + "/w.bin"; e.keygen("send", pk, vk, seed=1)
e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(w.send_instance(1))], wp); z = o.load_witness(wp); p = e.Prover(pk); p.set_witness(z)
for _ in range(5): p.prove_resident()
e.profile_enable(True)
for _ in range(N): p.prove_resident()
st = e.profile_report(); e.profile_enable(False); p.close()
print(" ".join("%s %.1f us" % (k, 1e3 * v["ms_total"] / v["count"]) for k, v in sorted(st.items()) if k.startswith(prefix)))
