#!/usr/bin/env python3
"""bench.py's set-up and warm-up (send witnesses generated on the host, stashed in HBM, proved from the stash) on a host whose cores are all busy with other work.
A round generates the same W statements' assignments again — they are deterministic, so any byte that differs from the first (quiet) round is a race in the witness
generator — stashes them and proves each from its stash and from the host buffer.   python tools/noisy_host.py [rounds] [hogs per core] [witnesses]"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6; per_core = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; W = int(sys.argv[3]) if len(sys.argv) > 3 else 55
import numpy as np
from blockmaze_amd import engine as e
import workload as w
from bench import read_witness
hx = lambda a: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in a]
tmp = tempfile.mkdtemp(); pk = os.path.join(tmp, "sendpk.txt"); e.keygen("send", pk, os.path.join(tmp, "sendvk.txt"), seed=11); prover = e.Prover(pk)
insts = [w.send_instance(i) for i in range(W)]; wp = os.path.join(tmp, "w.bin")
def witnesses():
    out = []
    for d in insts: e.witness_send(*hx(w.send_args(d)), wp); out.append(read_witness(wp))
    return out
quiet = witnesses()
cores = len(os.sched_getaffinity(0)); hogs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(int(cores * per_core))]
time.sleep(1.0); bad_witness = bad_stash = bad_host = 0
try:
    for r in range(rounds):
        zs = witnesses()
        for i, (a, b) in enumerate(zip(quiet, zs)):
            if not np.array_equal(a, b):
                bad_witness += 1; k = np.nonzero((a != b).any(axis=1))[0]; print("round %d: witness %d differs from the quiet one in %d variables, first %s" % (r, i, len(k), k[:8]))
        slots = []
        for z in zs: prover.set_witness(z); slots.append(prover.stash_witness())
        for i, s in enumerate(slots):
            try: prover.prove_stashed(s)
            except Exception as ex: bad_stash += 1; print("round %d: stash %d: %s" % (r, i, ex))
            try: prover.prove(zs[i])
            except Exception as ex: bad_host += 1; print("round %d: host buffer %d: %s" % (r, i, ex))
        prover.drop_stash()
finally:
    for h in hogs: h.kill()
    for h in hogs: h.wait()
print("%d rounds of %d statements beside %d busy processes on %d cores: %d witnesses differ, %d proofs from a stash failed, %d from the host buffer failed" % (rounds, W, len(hogs), cores, bad_witness, bad_stash, bad_host))
