#!/usr/bin/env python3
"""What bounds the rate with several proofs in flight?  K prover objects (sharing one key's tables), one host thread each, resident witnesses; optionally with parts of the
pipeline dropped (ZK_DEBUG_SKIP=w / h / n: witness MSMs / H query / transforms — the proofs are wrong then, only the timing means something; the switch exists
only in a diagnostic build of the library: `make -C blockmaze_amd/csrc clean && make -C blockmaze_amd/csrc HOOKS=1`, and rebuild without HOOKS afterwards).
    python tools/inflight_probe.py [K=3] [proofs per thread=60]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from blockmaze_amd import engine as e
import workload as w
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3; per = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hx = lambda args: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in args]
tmp = tempfile.mkdtemp(); pk, vk, wit = tmp + "/pk.txt", tmp + "/vk.txt", tmp + "/w.bin"; e.keygen("send", pk, vk, seed=7); e.witness_send(*hx(w.send_args(w.send_instance(1))), wit)
b = open(wit, "rb").read(); k = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); z = np.frombuffer(b, dtype=np.uint64, count=4 * k, offset=8).reshape(k, 4).copy()
p0 = e.Prover(pk); provers = [p0] + [p0.clone() for _ in range(K - 1)]
def run(p, n):
    for _ in range(n):
        try: p.prove_resident()
        except e.ZkGpuError: pass
for p in provers: p.set_witness(z); run(p, 3)
if os.environ.get("ZK_PROBE_START"): time.sleep(max(0.0, float(os.environ["ZK_PROBE_START"]) - time.time()))   # several processes on one GPU: start the timed part together
ths = [threading.Thread(target=run, args=(p, per)) for p in provers]; t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0; t0 = time.perf_counter(); run(p0, 30); one = (time.perf_counter() - t0) / 30
print("skip=%-4s K=%d: %.3f ms per proof in flight (%.0f /s) | one at a time %.3f ms" % (os.environ.get("ZK_DEBUG_SKIP", "-"), K, 1e3 * dt / (K * per), K * per / dt, 1e3 * one))
