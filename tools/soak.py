#!/usr/bin/env python3
"""Soak test: N send proofs from K threads through the cgo symbol genSendproof (pool of provers per key), every proof checked by the batched GPU verifier
and a sample by the host verifier.  python tools/soak.py [proofs_per_thread] [threads]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 250; K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ.setdefault("ZK_PROVERS_PER_KEY", str(K))
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp(); e.keygen("send", os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"), seed=99); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
insts = [w.send_instance(1000 + i) for i in range(32)]; out = [[] for _ in range(K)]
def worker(k):
    for i in range(per): d = insts[(7 * k + i) % len(insts)]; out[k].append((d, zk.GenSendProof(*w.send_args(d))))
ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]; t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
dt = time.time() - t0; allp = [x for o in out for x in o]
proofs = [p for _, p in allp]; ins = [w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]]) for d, _ in allp]
ok = e.verify_batch(os.path.join(tmp, "sendvk.txt"), proofs, ins); bad = [i for i, v in enumerate(ok) if not v]
host = all(e.verify(os.path.join(tmp, "sendvk.txt"), proofs[i], ins[i]) for i in range(0, len(proofs), max(1, len(proofs) // 20)))
print("%d proofs from %d threads in %.2f s = %.1f proofs/s through genSendproof; batched verifier rejected %d; host verifier sample %s; distinct proofs %d" % (len(proofs), K, dt, len(proofs) / dt, len(bad), "ok" if host else "FAILED", len(set(proofs))))
sys.exit(1 if bad or not host else 0)
