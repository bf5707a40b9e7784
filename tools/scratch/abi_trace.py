import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp(); e.keygen("send", os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"), seed=3); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
args = [w.send_args(w.send_instance(i)) for i in range(16)]
for i in range(40): zk.GenSendProof(*args[i % 16])
ts = []
for i in range(100): t0 = time.perf_counter(); zk.GenSendProof(*args[i % 16]); ts.append(1e3 * (time.perf_counter() - t0))
ts.sort(); print("genSendproof one caller: mean %.4f p50 %.4f p10 %.4f p90 %.4f ms" % (sum(ts) / len(ts), ts[50], ts[10], ts[90]), file=sys.stderr)
