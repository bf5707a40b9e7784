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
def sched():
    """per thread of this process: (ms on a CPU, ms runnable but waiting for one, time slices) from /proc/self/task/*/schedstat; and the cgroup's throttling counters"""
    out = {}
    for t in os.listdir("/proc/self/task"):
        try: a, b, c = open("/proc/self/task/%s/schedstat" % t).read().split(); out[int(t)] = (int(a) / 1e6, int(b) / 1e6, int(c))
        except Exception: pass
    return out
def throttle():
    try: return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("nr_periods", "nr_throttled", "throttled_usec")}
    except Exception: return {}
import threading
def vmstat():
    keys = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated", "pgmigrate_success", "thp_fault_alloc", "thp_collapse_alloc", "compact_stall", "pgfault", "pgmajfault")
    try: return {k: int(v) for k, v in (l.split() for l in open("/proc/vmstat")) if k in keys}      # (host-wide counters)
    except Exception: return {}
def faults(): f = open("/proc/self/stat").read().rsplit(")", 1)[1].split(); return int(f[7]) + int(f[9])       # minor + major faults of this process
main_tid = threading.get_native_id(); s0 = sched(); th0 = throttle(); vm0 = vmstat(); flt = []
ts = []; parts = []; waits = []
for i in range(N):
    w0 = int(open("/proc/self/task/%d/schedstat" % main_tid).read().split()[1]); f0 = faults()
    t0 = time.perf_counter(); p.prove(zs[i % 16]); ts.append(1e3 * (time.perf_counter() - t0)); parts.append(p.timings())
    waits.append((int(open("/proc/self/task/%d/schedstat" % main_tid).read().split()[1]) - w0) / 1e6); flt.append(faults() - f0)
s1 = sched(); th1 = throttle(); vm1 = vmstat()
s = sorted(ts); pct = lambda q: s[min(len(s) - 1, int(q * len(s)))]
print("%d steps: mean %.3f ms, min %.3f, p10 %.3f, median %.3f, p90 %.3f, p99 %.3f, max %.3f; steps above 1.5 x median: %d (they add %.3f ms to the mean)" % (N, sum(ts) / N, s[0], pct(0.1), pct(0.5), pct(0.9), pct(0.99), s[-1], sum(1 for t in ts if t > 1.5 * pct(0.5)), sum(t - pct(0.5) for t in ts if t > 1.5 * pct(0.5)) / N))
med = lambda k: sorted(d[k] for d in parts)[len(parts) // 2]
print("  medians of the prover's own clocks: " + ", ".join("%s %.3f" % (k, med(k)) for k in ("upload_ms", "enqueue_ms", "device_ms", "finish_ms", "total_ms")) + "; cpus allowed %d; AnonHugePages of this process %s kB" % (len(os.sched_getaffinity(0)), next((l.split()[1] for l in open("/proc/self/smaps_rollup") if l.startswith("AnonHugePages")), "?")))
slow = [(t, d) for t, d in zip(ts, parts) if t > 1.5 * pct(0.5)]
if slow: print("  the slow steps' own clocks (step: upload / device / finish ms): " + "; ".join("%.2f: %.2f / %.2f / %.2f" % (t, d["upload_ms"], d["device_ms"], d["finish_ms"]) for t, d in slow[:12]))
print("  the calling thread waited for a CPU (schedstat run delay): %.3f ms in all, %.3f ms of it inside the %d slow steps (%s)" % (sum(waits), sum(x for t, x in zip(ts, waits) if t > 1.5 * pct(0.5)), len(slow), ", ".join("%.2f" % x for t, x in zip(ts, waits) if t > 1.5 * pct(0.5))[:200]))
print("  per thread over the run (ms on a CPU / ms waiting for one / slices): " + "; ".join("%s%.0f / %.1f / %d" % ("main " if t == main_tid else "", s1[t][0] - s0.get(t, (0, 0, 0))[0], s1[t][1] - s0.get(t, (0, 0, 0))[1], s1[t][2] - s0.get(t, (0, 0, 0))[2]) for t in sorted(s1) if s1[t][0] - s0.get(t, (0, 0, 0))[0] > 1.0))
if th1: print("  cgroup cpu.stat over the run: " + ", ".join("%s +%d" % (k, th1[k] - th0.get(k, 0)) for k in th1) + "; cpu.max: " + open("/sys/fs/cgroup/cpu.max").read().strip())
print("  page faults of this process per step: median %d, in the slow steps %s; host-wide /proc/vmstat over the run: %s; kernel.numa_balancing = %s" % (sorted(flt)[len(flt) // 2], [x for t, x in zip(ts, flt) if t > 1.5 * pct(0.5)][:12], ", ".join("%s +%d" % (k, vm1[k] - vm0[k]) for k in vm1), open("/proc/sys/kernel/numa_balancing").read().strip() if os.path.exists("/proc/sys/kernel/numa_balancing") else "?"))
print("RESULT median_ms %.4f p10_ms %.4f device_ms %.4f upload_ms %.4f" % (pct(0.5), pct(0.1), med("device_ms"), med("upload_ms")))
