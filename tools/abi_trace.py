#!/usr/bin/env python3
"""Per-call breakdown of genSendproof (ZK_TRACE_TIMES): acquire / witness generation / hand-over / prove / hex, for a single caller."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ["ZK_TRACE_TIMES"] = "1"
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp(); e.keygen("send", tmp + "/sendpk.txt", tmp + "/sendvk.txt", seed=7); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
for i in range(12): zk.GenSendProof(*w.send_args(w.send_instance(i % 3)))
