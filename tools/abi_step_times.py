#!/usr/bin/env python3
"""Distribution of the per-call time of genSendproof from ONE caller (witness generation + hand-over + proof + hex), distinct instances.  python tools/abi_step_times.py [N]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
import workload as w
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tmp = tempfile.mkdtemp(); e.keygen("send", tmp + "/sendpk.txt", tmp + "/sendvk.txt", seed=7); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
args = [w.send_args(w.send_instance(i)) for i in range(16)]; keep = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
for i in range(8): zk.GenSendProof(*args[i % 16])
ts = []
for i in range(N):
    t0 = time.perf_counter(); zk.GenSendProof(*args[i % 16]); ts.append(1e3 * (time.perf_counter() - t0))
os.dup2(keep, 1); s = sorted(ts); pct = lambda q: s[min(len(s) - 1, int(q * len(s)))]
print("%d calls: mean %.3f ms, min %.3f, p10 %.3f, median %.3f, p90 %.3f, p99 %.3f, max %.3f" % (N, sum(ts) / N, s[0], pct(0.1), pct(0.5), pct(0.9), pct(0.99), s[-1]))
