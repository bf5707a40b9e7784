#!/usr/bin/env python3
"""Distribution of the per-proof time of a host-buffer send proof (the quantity bench.py averages): percentiles over N steps with distinct witnesses.  python tools/step_times.py [N]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import workload as w
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
def place(a):
    """STEP_PAGES=huge / small: the assignment in an anonymous mapping advised to use / not to use transparent huge pages (default: numpy's own allocation)"""
    mode = os.environ.get("STEP_PAGES", "")
    if mode not in ("huge", "small"): return a
    import mmap, numpy as np
    size = (a.nbytes + (4 << 20)) & ~((2 << 20) - 1); m = mmap.mmap(-1, size); m.madvise(mmap.MADV_HUGEPAGE if mode == "huge" else mmap.MADV_NOHUGEPAGE)
    addr = np.frombuffer(m, dtype=np.uint8).ctypes.data; off = (-addr) % (2 << 20)                    # start on a 2 MB boundary
    b = np.frombuffer(m, dtype=np.uint64, count=a.size, offset=off).reshape(a.shape); b[...] = a; keep.append(m); return b
keep = []
tmp = tempfile.mkdtemp(); pk, vk = os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt"); e.keygen("send", pk, vk, seed=1); p = e.Prover(pk); zs = []
for i in range(16):
    d = w.send_instance(i); wp = os.path.join(tmp, "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); zs.append(place(o.load_witness(wp)))
for i in range(10): p.prove(zs[i % 16])
ts = []; parts = []
for i in range(N):
    t0 = time.perf_counter(); p.prove(zs[i % 16]); ts.append(1e3 * (time.perf_counter() - t0)); parts.append(p.timings())
s = sorted(ts); pct = lambda q: s[min(len(s) - 1, int(q * len(s)))]
print("%d steps: mean %.3f ms, min %.3f, p10 %.3f, median %.3f, p90 %.3f, p99 %.3f, max %.3f; steps above 1.5 x median: %d (they add %.3f ms to the mean)" % (N, sum(ts) / N, s[0], pct(0.1), pct(0.5), pct(0.9), pct(0.99), s[-1], sum(1 for t in ts if t > 1.5 * pct(0.5)), sum(t - pct(0.5) for t in ts if t > 1.5 * pct(0.5)) / N))
med = lambda k: sorted(d[k] for d in parts)[len(parts) // 2]
print("  medians of the prover's own clocks: " + ", ".join("%s %.3f" % (k, med(k)) for k in ("upload_ms", "enqueue_ms", "device_ms", "finish_ms", "total_ms")) + "; cpus allowed %d; AnonHugePages of this process %s kB" % (len(os.sched_getaffinity(0)), next((l.split()[1] for l in open("/proc/self/smaps_rollup") if l.startswith("AnonHugePages")), "?")))
slow = [(t, d) for t, d in zip(ts, parts) if t > 1.5 * pct(0.5)]
if slow: print("  the slow steps' own clocks (step: upload / device / finish ms): " + "; ".join("%.2f: %.2f / %.2f / %.2f" % (t, d["upload_ms"], d["device_ms"], d["finish_ms"]) for t, d in slow[:12]))
print("RESULT median_ms %.4f p10_ms %.4f device_ms %.4f upload_ms %.4f" % (pct(0.5), pct(0.1), med("device_ms"), med("upload_ms")))
