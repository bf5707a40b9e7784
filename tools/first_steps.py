import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
tmp = tempfile.mkdtemp(); pk, vk = os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"); e.keygen("send", pk, vk, seed=1); p = e.Prover(pk); zs = []
for i in range(8):
    d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(o.load_witness(wp))
slots = []
for z in zs: p.set_witness(z); slots.append(p.stash_witness())
ts = []
for i in range(60):
    t0 = time.perf_counter(); p.prove_stashed(slots[i % 8]); ts.append(1e3 * (time.perf_counter() - t0))
print("first 60 steps of a fresh prover (ms):", " ".join("%.2f" % t for t in ts))
time.sleep(2.0); ts = []
for i in range(20):
    t0 = time.perf_counter(); p.prove_stashed(slots[i % 8]); ts.append(1e3 * (time.perf_counter() - t0))
print("after 2 s idle:", " ".join("%.2f" % t for t in ts))
