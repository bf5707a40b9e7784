#!/usr/bin/env python3
"""Which mint statements send a witness MSM to the general path?  N genMintproof calls from ONE thread over 64 statements, the repeat counter read after each.
(Round 6: none in 40,000 — the 0 to 2 repeats of a 57,600-proof mixed soak need the four concurrent callers: the same statements, other arrival orders of the sort's atomics.)  python tools/mint_hunt.py [N] [threads]"""
import os, sys, tempfile, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from blockmaze_amd import engine as e
import workload as w
tmp = os.environ.get("ZK_HUNT_DIR") or tempfile.mkdtemp(); os.makedirs(tmp, exist_ok=True); e.keygen("mint", os.path.join(tmp, "mintpk.txt"), os.path.join(tmp, "mintvk.txt"), seed=0xB10C4A2E + 4); os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000; K = 64
ms = [w.mint_instance(i) for i in range(K)]; hits = {}
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1; kind = sys.argv[3] if len(sys.argv) > 3 else "mint"
import threading
keep = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)
before = e.general_path_repeats(); lock = threading.Lock()
def worker(t):
    last = e.general_path_repeats()
    for i in range(t, N, T):
        m = ms[i % K]; zk.GenMintProof(*w.mint_args(m)); now = e.general_path_repeats()
        if now != last:
            with lock: hits[i % K] = hits.get(i % K, 0) + 1
            last = now
ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for t in ths: t.start()
for t in ths: t.join()
os.dup2(keep, 1)
print("%d mint proofs over %d statements from %d thread(s): %d repeats; calls that saw the counter move, by statement: %s" % (N, K, T, e.general_path_repeats() - before, hits))
