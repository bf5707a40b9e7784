"""Device time of the batched verifier over batch sizes (HIP-event stage time of acc + K9 or the lane kernel).  ZK_VERIFY_WAVE_MAX=n: workgroup kernel up to n proofs.  python tools/verify_sweep.py"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
d = os.path.join(ROOT, "tests", "golden", "groth16_small"); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
inputs = o.from_arr(z[:meta["n_inputs"]]); e.init()
for n in (1, 64, 256, 512, 1024, 2048, 4096, 8192, 16384):
    pr = [meta["proof"]] * n; ins = [inputs] * n; r = e.verify_batch(vk, pr, ins); assert all(r)
    e.profile_enable(True); e.verify_batch(vk, pr, ins); e.verify_batch(vk, pr, ins); st = e.profile_report(); e.profile_enable(False); dev = st["verify.batch"]["ms_total"] / st["verify.batch"]["count"]
    print("ZK_VERIFY_WAVE_MAX=%s n = %6d: device %8.2f ms = %9.0f proofs/s" % (os.environ.get("ZK_VERIFY_WAVE_MAX", "default"), n, dev, n / dev * 1e3), flush=True)
