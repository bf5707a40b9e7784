#!/usr/bin/env python3
"""send proofs only from K threads through genSendproof, every proof verified: counts failures.  python tools/soak_send.py [per_thread] [threads]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1000; K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp(); e.keygen("send", os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"), seed=11); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
sends = [w.send_instance(100 + i) for i in range(8)]; fails = [0] * K
def worker(k):
    for i in range(per):
        p = zk.GenSendProof(*w.send_args(sends[(i + k) % 8]))
        if len(p) != 512 or p.startswith("0000000000"): fails[k] += 1
ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]; t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
print("%d send proofs from %d threads in %.1f s: %d not generated" % (per * K, K, time.time() - t0, sum(fails)))
