#!/usr/bin/env python3
"""Per-stream timeline of one proof from a rocprofv3 --kernel-trace CSV.

    python tools/timeline.py <kernel_trace.csv> [proof_number]

Splits the trace into proofs at each k_r1cs_rows launch (first kernel of the critical chain), then prints for the chosen
proof every stream's busy intervals, the kernels on the critical path and the idle gaps of the main stream.
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
def col(r, *names):
    for n in names:
        if n in r: return r[n]
    raise KeyError(names)
ev = sorted(((int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name"), col(r, "Stream_Id", "Queue_Id")) for r in rows), key=lambda t: t[0])
starts = []
for i, e in enumerate(ev):
    if "k_r1cs_rows" in e[2] and (not starts or e[0] - ev[starts[-1]][0] > 300000): starts.append(i)       # (row kernels less than 0.3 ms apart belong to one proof; proofs follow each other in less than 1 ms since round 4)
k = back if back < len(starts) - 1 else len(starts) - 2          # argument = proof number from the start of the trace
lo = ev[starts[k]][0] - 100000; hi = ev[starts[k + 1]][0] - 100000
sel = [e for e in ev if lo <= e[0] < hi]
t0 = min(e[0] for e in sel); tend = max(e[1] for e in sel)
print("proof window: %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(sel)))
by = collections.defaultdict(list)
for e in sel: by[e[3]].append(e)
def short(n): return n.replace("zk::", "").split("(")[0][:60]
for s, lst in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e[1] - e[0] for e in lst)
    print("\nstream %s: %d kernels, busy %.3f ms, span %.3f..%.3f ms" % (s, len(lst), busy / 1e6, (lst[0][0] - t0) / 1e6, (lst[-1][1] - t0) / 1e6))
    last = None
    for e in lst:
        gap = (e[0] - last) / 1e3 if last else 0.0
        if (e[1] - e[0]) > 20000 or gap > 20: print("   %8.3f +%7.1f us (gap %6.1f us)  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e3, gap, short(e[2])))
        last = e[1]
