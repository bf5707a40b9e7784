#!/usr/bin/env python3
"""Where the hand-over of a host-buffer assignment spends its time on the host (ZK_TRACE_TIMES lines of Prover::set_witness): medians over N proofs that cycle through W distinct
assignments (W = 55 is bench.py's 400 MB, nothing stays in the last-level cache).  python tools/handover_trace.py [N] [W]   — honours ZK_SCAN_THREADS / ZK_SPIN_US"""
import os, re, subprocess, sys
if os.environ.get("HANDOVER_CHILD"):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tempfile, time
    from blockmaze_amd import engine as e
    from oracle import pyoracle as o
    import workload as w
    N, W = int(sys.argv[1]), int(sys.argv[2])
    tmp = tempfile.mkdtemp(); pk = os.path.join(tmp, "sendpk.txt"); e.keygen("send", pk, os.path.join(tmp, "sendvk.txt"), seed=1); p = e.Prover(pk); zs = []
    for i in range(W):
        d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(o.load_witness(wp))
    for i in range(10): p.prove(zs[i % W])
    sys.stderr.write("trace-begin\n"); sys.stderr.flush(); ts = []
    for i in range(N):
        t0 = time.perf_counter(); p.prove(zs[i % W]); ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort(); print("step median %.3f p10 %.3f p90 %.3f" % (ts[len(ts) // 2], ts[len(ts) // 10], ts[9 * len(ts) // 10])); sys.exit(0)
N = sys.argv[1] if len(sys.argv) > 1 else "400"; W = sys.argv[2] if len(sys.argv) > 2 else "55"
r = subprocess.run([sys.executable, os.path.abspath(__file__), N, W], env=dict(os.environ, HANDOVER_CHILD="1", ZK_TRACE_TIMES="1"), capture_output=True, text=True)
err = r.stderr.split("trace-begin", 1)[-1]; rows = [tuple(float(x) for x in m.groups()) for m in re.finditer(r"trace-handover-host: threads (\d+) post ([\d.]+) own scan ([\d.]+) join ([\d.]+) copy \+ expand calls ([\d.]+)", err)]
med = lambda k: sorted(x[k] for x in rows)[len(rows) // 2] if rows else float("nan")
print("%s; hand-over medians over %d proofs: threads %d, post %.3f, own scan %.3f, join %.3f, copy + expand calls %.3f ms" % (r.stdout.strip(), len(rows), int(med(0)) if rows else 0, med(1), med(2), med(3), med(4)))
