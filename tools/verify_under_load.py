#!/usr/bin/env python3
"""verifySendproof latency of one caller while K threads keep the GPU busy with genSendproof (ZK_VERIFY_GPU_MIN=1000000 in the environment: the host verifier).  python tools/verify_under_load.py [K]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
import workload as w
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tmp = tempfile.mkdtemp(); e.keygen("send", tmp + "/sendpk.txt", tmp + "/sendvk.txt", seed=7); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
insts = [w.send_instance(i) for i in range(8)]; keep = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
d0 = insts[0]; proof = zk.GenSendProof(*w.send_args(d0)).decode() if isinstance(zk.GenSendProof(*w.send_args(d0)), bytes) else zk.GenSendProof(*w.send_args(d0)); vargs = (proof if isinstance(proof, str) else proof.decode(), d0["cmtA_old"], d0["sn_old"], d0["cmtS"], d0["cmtA"])
assert zk.VerifySendProof(*vargs)
def lat(n=150):
    ts = []
    for _ in range(n): t0 = time.perf_counter(); ok = zk.VerifySendProof(*vargs); ts.append(1e3 * (time.perf_counter() - t0)); assert ok
    s = sorted(ts); return "median %.2f ms, p90 %.2f, max %.2f" % (s[len(s) // 2], s[int(0.9 * len(s))], s[-1])
idle = lat(); stop = False; count = [0] * K
def load(k):
    i = 0
    while not stop: zk.GenSendProof(*w.send_args(insts[(i + k) % 8])); i += 1; count[k] = i
ths = [threading.Thread(target=load, args=(k,)) for k in range(K)]
for t in ths: t.start()
time.sleep(0.5); t0 = time.perf_counter(); c0 = sum(count); busy = lat(); rate = (sum(count) - c0) / (time.perf_counter() - t0); stop = True
for t in ths: t.join()
os.dup2(keep, 1); print("verifySendproof (ZK_VERIFY_GPU_MIN=%s): GPU idle: %s | %d provers busy (%.0f proofs/s meanwhile): %s" % (os.environ.get("ZK_VERIFY_GPU_MIN", "default"), idle, K, rate, busy))
