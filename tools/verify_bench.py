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
t0 = time.perf_counter(); ok = [e.verify(vk, *base[i % 4]) for i in range(8)]; th = (time.perf_counter() - t0) / 8; assert all(ok)
print("host verifier (includes re-reading the vk file): %.2f ms per proof" % (1e3 * th))
e.verify_batch(vk, [base[0][0]], [base[0][1]])
for n in (1, 64, 512, 4096, 16384, 65536):
    proofs = [base[i % 4][0] for i in range(n)]; ins = [base[i % 4][1] for i in range(n)]
    t0 = time.perf_counter(); r = e.verify_batch(vk, proofs, ins); t = time.perf_counter() - t0; assert all(r)
    e.profile_enable(True); e.verify_batch(vk, proofs, ins); st = e.profile_report(); e.profile_enable(False); dev = st["verify.batch"]["ms_total"]
    print("n = %6d: %9.2f ms end to end (hex parsing and marshalling in Python included), device %9.2f ms = %8.0f proofs/s" % (n, 1e3 * t, dev, n / dev * 1e3))
