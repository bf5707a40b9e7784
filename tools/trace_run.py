import os, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ["ZK_TRACE_TIMES"] = "1"
import numpy as np
from blockmaze_amd import engine as e
import workload as w
hx = lambda args: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in args]
tmp = tempfile.mkdtemp(); pk, vk, wit = tmp + "/pk.txt", tmp + "/vk.txt", tmp + "/w.bin"
e.keygen("send", pk, vk, seed=7); e.witness_send(*hx(w.send_args(w.send_instance(1))), wit)
b = open(wit, "rb").read(); k = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); z = np.frombuffer(b, dtype=np.uint64, count=4 * k, offset=8).reshape(k, 4).copy()
p = e.Prover(pk); p.set_witness(z)
for _ in range(12): p.prove_resident()
