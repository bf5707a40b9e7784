import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
d = os.path.join(ROOT, "tests", "golden", "groth16_small"); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
inputs = o.from_arr(z[:meta["n_inputs"]]); e.init()
for n in (1, 1, 1, 1, 1, 1, 1, 1, 64, 64, 64, 64):
    t0 = time.perf_counter(); r = e.verify_batch(vk, [meta["proof"]] * n, [inputs] * n); assert all(r)
print("done")
