#!/usr/bin/env python3
"""How often does a 20-step window — what the driver times — contain a slow step?  One process, TRIALS windows of (idle, 35 untimed proofs, 5 warm-up, 20 timed) on
statements resident in HBM.  python tools/window_probe.py [trials]   (PROBE_TORCH=1: import torch first, as bench.py does; PROBE_IDLE_S: the pause between windows)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
if os.environ.get("PROBE_TORCH") == "1":
    import torch; torch.cuda.set_device(0)
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
import gc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 40; idle = float(os.environ.get("PROBE_IDLE_S", "0.3"))
tmp = tempfile.mkdtemp(); pk, vk = os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"); e.keygen("send", pk, vk, seed=1); p = e.Prover(pk); zs = []
for i in range(8):
    d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(o.load_witness(wp))
slots = []
for z in zs: p.set_witness(z); slots.append(p.stash_witness())
gc.collect(); gc.disable(); means = []; slow = []; worst = []
for t in range(T):
    time.sleep(idle)
    for i in range(40): p.prove_stashed(slots[i % 8])
    ts = []
    for i in range(20):
        t0 = time.perf_counter(); p.prove_stashed(slots[i % 8]); ts.append(1e3 * (time.perf_counter() - t0))
    med = sorted(ts)[10]; means.append(sum(ts) / 20); slow.append(sum(1 for x in ts if x > 1.5 * med)); worst.append(max(ts))
print("%d windows of 20 steps: window means min %.3f median %.3f max %.3f ms; windows with a slow step (> 1.5 x the window's median): %d; slow steps in all: %d of %d; worst steps: %s" % (T, min(means), sorted(means)[T // 2], max(means), sum(1 for s in slow if s), sum(slow), 20 * T, " ".join("%.1f" % x for x in sorted(worst)[-6:])))
