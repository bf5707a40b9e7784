#!/usr/bin/env python3
"""Throughput of the batched GPU verifier (kernel K9) against the host verifier, send circuit.  python tools/verify_bench.py"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
tmp = tempfile.mkdtemp(); pk, vk = os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"); e.keygen("send", pk, vk, seed=1); p = e.Prover(pk); base = []
for i in range(4):
    d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp)
    base.append((p.prove(o.load_witness(wp)), w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]])))
p.close()
if len(sys.argv) > 1 and sys.argv[1] == "abi":   # the cgo symbol verifySendproof, one proof per call, from 1 and from 8 threads (ZK_VERIFY_GPU_MIN=1000000 in the environment: the host verifier)
    import threading
    os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk(); d0 = w.send_instance(0); args = (base[0][0], d0["cmtA_old"], d0["sn_old"], d0["cmtS"], d0["cmtA"]); devnull = os.open(os.devnull, os.O_WRONLY); keep = os.dup(1); os.dup2(devnull, 1)   # (the symbol prints a line per call, like the reference)
    assert zk.VerifySendProof(*args) and not zk.VerifySendProof(base[1][0], *args[1:])
    res = []
    for K in (1, 8):
        n = 200
        def work():
            for _ in range(n): zk.VerifySendProof(*args)
        ths = [threading.Thread(target=work) for _ in range(K)]; t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0; res.append("%d thread(s): %.2f ms per call, %.0f verifications/s" % (K, 1e3 * dt / n, K * n / dt))
    os.dup2(keep, 1); print("verifySendproof (ZK_VERIFY_GPU_MIN=%s): " % os.environ.get("ZK_VERIFY_GPU_MIN", "default") + "; ".join(res)); sys.exit(0)
t0 = time.perf_counter(); ok = [e.verify(vk, *base[i % 4]) for i in range(8)]; th = (time.perf_counter() - t0) / 8; assert all(ok)
print("host verifier (includes re-reading the vk file): %.2f ms per proof" % (1e3 * th))
e.verify_batch(vk, [base[0][0]], [base[0][1]])
for n in (1, 64, 512, 4096, 16384, 65536):
    proofs = [base[i % 4][0] for i in range(n)]; ins = [base[i % 4][1] for i in range(n)]
    t0 = time.perf_counter(); r = e.verify_batch(vk, proofs, ins); t = time.perf_counter() - t0; assert all(r)
    e.profile_enable(True); e.verify_batch(vk, proofs, ins); st = e.profile_report(); e.profile_enable(False); dev = st["verify.batch"]["ms_total"]
    print("n = %6d: %9.2f ms end to end (hex parsing and marshalling in Python included), device %9.2f ms = %8.0f proofs/s" % (n, 1e3 * t, dev, n / dev * 1e3))
