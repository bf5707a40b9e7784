import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
d = os.path.join(ROOT, "tests", "golden", "groth16_small"); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
inputs = o.from_arr(z[:meta["n_inputs"]]); e.init()
print("fetch", os.environ.get("ZK_VERIFY_FETCH"), "trace:", e.verify_trace(vk, meta["proof"], inputs, 1))
