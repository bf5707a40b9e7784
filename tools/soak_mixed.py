#!/usr/bin/env python3
"""Soak test of BASELINE.json configs[3] at volume: mint, deposit (depth-8 Merkle tree), redeem and send proofs interleaved from K threads through the cgo symbols
(four resident keys, a pool of provers each), every proof checked through the matching verifyXproof symbol.  python tools/soak_mixed.py [rounds_per_thread] [threads]"""
import os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
per = int(sys.argv[1]) if len(sys.argv) > 1 else 50; K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp()
for kind in ("send", "mint", "redeem", "deposit"): e.keygen(kind, os.path.join(tmp, kind + "pk.txt"), os.path.join(tmp, kind + "vk.txt"), seed=0xB10C4A2E + len(kind))
os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
mints = [w.mint_instance(i) for i in range(8)]; redeems = [w.mint_instance(i, redeem=True) for i in range(8)]; sends = [w.send_instance(100 + i) for i in range(8)]; deps = [w.deposit_instance(i) for i in range(4)]
out = [[] for _ in range(K)]
def worker(k):
    for i in range(per):
        j = 3 * k + i
        m = mints[j % 8]; out[k].append(("mint", m, zk.GenMintProof(*w.mint_args(m))))
        d = deps[j % 4]; out[k].append(("deposit", d, zk.GenDepositProof(*w.deposit_args(d), d["leaves"], d["rt"], d["sk"])))
        r = redeems[j % 8]; out[k].append(("redeem", r, zk.GenRedeemProof(*w.mint_args(r))))
        s = sends[j % 8]; out[k].append(("send", s, zk.GenSendProof(*w.send_args(s))))
ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]; t0 = time.time()
for t in ths: t.start()
for t in ths: t.join()
dt = time.time() - t0; allp = [x for o_ in out for x in o_]
def check(kind, d, p):
    if len(p) != 512 or p.startswith("0000000000"): return False
    if kind == "mint": return zk.VerifyMintProof(p, d["cmtA_old"], d["sn_old"], d["cmtA"], d["value_s"]) and not zk.VerifyMintProof(p, d["cmtA_old"], d["sn_old"], d["cmtA"], d["value_s"] + 1)
    if kind == "redeem": return zk.VerifyRedeemProof(p, d["cmtA_old"], d["sn_old"], d["cmtA"], d["value_s"])
    if kind == "send": return zk.VerifySendProof(p, d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"])
    return zk.VerifyDepositProof(p, d["rt"], d["pk_recv"], d["cmtB_old"], d["sn_old"], d["cmtB"], d["sn_s"])
t1 = time.time(); bad = [i for i, (kind, d, p) in enumerate(allp) if not check(kind, d, p)]; tv = time.time() - t1
print("%d proofs (mint, deposit, redeem, send interleaved) from %d threads in %.2f s = %.1f proofs/s through the cgo symbols, key loads included; rejected by the verify symbols: %d; distinct proofs %d; %d verifications in %.2f s" % (len(allp), K, dt, len(allp) / dt, len(bad), len(set(p for _, _, p in allp)), len(allp) + len(allp) // 4, tv))
for i in bad[:8]:
    kind, d, p = allp[i]; print("  rejected: #%d %s proof %s... (%s)" % (i, kind, p[:16], "the failure sentinel: no proof was generated" if p.startswith("0000000000") or len(p) != 512 else "a proof the verifier refuses"))
print("MSMs repeated on the general path (a fast path raised its flag): %d" % e.general_path_repeats())
sys.exit(1 if bad else 0)
