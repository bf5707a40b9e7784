#!/usr/bin/env python3
"""When the host's share of a proof's tail is done against when the device is (the `trace:` lines of Prover::prove_resident under ZK_TRACE_TIMES): medians over N host-buffer send
proofs, times in ms from the start of the prover call.  python tools/trace_tail.py [N]"""
import os, re, subprocess, sys
if os.environ.get("TRACE_TAIL_CHILD"):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tempfile
    from blockmaze_amd import engine as e
    from oracle import pyoracle as o
    import workload as w
    N = int(sys.argv[1]); tmp = tempfile.mkdtemp(); pk = os.path.join(tmp, "sendpk.txt"); e.keygen("send", pk, os.path.join(tmp, "sendvk.txt"), seed=1); p = e.Prover(pk); zs = []
    for i in range(8):
        d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(o.load_witness(wp))
    for i in range(10): p.prove(zs[i % 8])
    sys.stderr.write("trace-begin\n"); sys.stderr.flush()
    for i in range(N): p.prove(zs[i % 8])
    sys.exit(0)
N = sys.argv[1] if len(sys.argv) > 1 else "200"
r = subprocess.run([sys.executable, os.path.abspath(__file__), N], env=dict(os.environ, TRACE_TAIL_CHILD="1", ZK_TRACE_TIMES="1"), capture_output=True, text=True)
err = r.stderr.split("trace-begin", 1)[-1]
rows = [tuple(float(x) for x in m.groups()) for m in re.finditer(r"trace: enqueue ([\d.]+) A ([\d.]+) L ([\d.]+) B1 ([\d.]+) B2 ([\d.]+) sync ([\d.]+)", err)]
med = lambda k: sorted(x[k] for x in rows)[len(rows) // 2] if rows else float("nan")
print("%d proofs, medians from the start of the call: enqueue done %.3f, A settled and s*A formed %.3f, L %.3f, B1 settled and r*B1 formed %.3f, B2 converted %.3f, device done %.3f ms; the host was ready %.3f ms before the device" % (len(rows), med(0), med(1), med(2), med(3), med(4), med(5), med(5) - med(4)))
