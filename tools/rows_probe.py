#!/usr/bin/env python3
"""Runs the R1CS row kernels of the send circuit alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
tmp = tempfile.mkdtemp(); rp, wp = os.path.join(tmp, "r.bin"), os.path.join(tmp, "w.bin"); e.circuit_export("send", rp); d = w.send_instance(1)
e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); z = o.load_witness(wp); cs = o.R1CS.load(rp)
r = e.R1cs(cs.n_inputs, cs.n_vars, cs.n_cons, cs.rowptr, cs.col, cs.coeff)
for _ in range(10): r.witness_map(z)
print("ok")
