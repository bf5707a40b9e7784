#!/usr/bin/env python3
"""Per-call breakdown of genSendproof (ZK_TRACE_TIMES): acquire / witness generation / hand-over / prove / hex, for a single caller."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ["ZK_TRACE_TIMES"] = "1"
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp(); e.keygen("send", tmp + "/sendpk.txt", tmp + "/sendvk.txt", seed=7); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
import time
insts = [w.send_instance(i) for i in range(3)]; args = [w.send_args(d) for d in insts]
for i in range(12):
    t0 = time.perf_counter(); zk.GenSendProof(*args[i % 3]); t1 = time.perf_counter(); sys.stderr.write("python: %.3f ms for the whole call (ctypes marshalling included)\n" % (1e3 * (t1 - t0)))
