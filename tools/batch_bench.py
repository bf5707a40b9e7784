#!/usr/bin/env python3
"""prove_batch against K provers in flight, send circuit: python tools/batch_bench.py [batch] [reps]   (env ZK_BATCH_LANES)"""
import os, sys, tempfile, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = tempfile.mkdtemp(); pk, vk = tmp + "/pk.txt", tmp + "/vk.txt"; e.keygen("send", pk, vk, seed=1); p = e.Prover(pk); zs = []
for i in range(16):
    d = w.send_instance(i); wp = tmp + "/w.bin"; e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(o.load_witness(wp))
batch = np.ascontiguousarray(np.stack([zs[i % 16] for i in range(B)])); p.prove_batch(batch); t0 = time.perf_counter()
for _ in range(reps): p.prove_batch(batch)
dt = time.perf_counter() - t0; print("prove_batch of %d (lanes %s): %.1f proofs/s, %.4f ms per proof" % (B, os.environ.get("ZK_BATCH_LANES", "default"), B * reps / dt, 1e3 * dt / (B * reps)))
K = int(os.environ.get("INFLIGHT", "6")); provers = [p] + [p.clone() for _ in range(K - 1)]; per = B * reps // K
for k, pv in enumerate(provers): pv.prove(zs[k % 16])
def worker(k):
    for i in range(per): provers[k].prove(zs[(i + k) % 16])
ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]; t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0; print("%d provers in flight: %.1f proofs/s" % (K, per * K / dt))
