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
def cpu_stat():
    """the cgroup's CPU time and throttling counters, and this process's CPU time: a pod with a CPU quota that its spinning helper threads exhaust is stopped for the rest of the period"""
    try: d = {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception: d = {}
    d["process_cpu_s"] = time.process_time(); return d
def cpu_report(a, b, dt):
    return "CPUs busy %.1f; cgroup: throttled %d of %d periods, %.1f ms (cpu.max %s)" % ((b["process_cpu_s"] - a["process_cpu_s"]) / dt, b.get("nr_throttled", 0) - a.get("nr_throttled", 0), b.get("nr_periods", 0) - a.get("nr_periods", 0), (b.get("throttled_usec", 0) - a.get("throttled_usec", 0)) / 1e3, open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "-")
batch = np.ascontiguousarray(np.stack([zs[i % 16] for i in range(B)])); p.prove_batch(batch); c0 = cpu_stat(); t0 = time.perf_counter()
for _ in range(reps): p.prove_batch(batch)
dt = time.perf_counter() - t0; print("prove_batch of %d (lanes %s): %.1f proofs/s, %.4f ms per proof; %s" % (B, os.environ.get("ZK_BATCH_LANES", "default"), B * reps / dt, 1e3 * dt / (B * reps), cpu_report(c0, cpu_stat(), dt)))
K = int(os.environ.get("INFLIGHT", "6")); provers = [p] + [p.clone() for _ in range(K - 1)]; per = B * reps // K
for k, pv in enumerate(provers): pv.prove(zs[k % 16])
def worker(k):
    for i in range(per): provers[k].prove(zs[(i + k) % 16])
ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]; c0 = cpu_stat(); t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0; print("%d provers in flight: %.1f proofs/s; %s" % (K, per * K / dt, cpu_report(c0, cpu_stat(), dt)))
