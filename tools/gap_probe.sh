#!/bin/bash
# Where a proof's wall time goes OUTSIDE its kernels: host time stamps of the prover (ZK_TRACE_TIMES, CLOCK_BOOTTIME) laid over rocprofv3's kernel and copy trace.
# bash tools/gap_probe.sh <tag>   (GPU box, repo root)
tag=${1:-r03}; root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
ZK_TRACE_TIMES=1 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/gap_${tag} -- python3 $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/gap_${tag}.log 2> $root/gpurun_out/gap_${tag}.err
cd $root
python3 - <<PY > gpurun_out/${tag}_gaps.txt
import csv, glob, re
kt = glob.glob("gpurun_out/gap_${tag}/**/*kernel_trace.csv", recursive=True)[0]; mc = glob.glob("gpurun_out/gap_${tag}/**/*memory_copy_trace.csv", recursive=True)
ev = [(int(r["Start_Timestamp"]) * 1e-6, int(r["End_Timestamp"]) * 1e-6, r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")) for r in csv.DictReader(open(kt))]
if mc: ev += [(int(r["Start_Timestamp"]) * 1e-6, int(r["End_Timestamp"]) * 1e-6, "copy " + r.get("Direction", "")) for r in csv.DictReader(open(mc[0]))]
ev.sort()
for line in open("gpurun_out/gap_${tag}.err"):
    m = re.search(r"trace: enqueue ([\d.]+) .* sync ([\d.]+) upload ([\d.]+) t1_boot ([\d.]+)", line)
    if not m: continue
    enq, sync, up, t1 = map(float, m.groups()); t0 = t1 - up
    mine = [e for e in ev if t0 - 0.05 <= e[0] <= t1 + sync + 0.05]
    if not mine: continue
    first_k = min(e[0] for e in mine if not e[2].startswith("copy")); last = max(e[1] for e in mine)
    print("proof: set_witness %.3f ms | t1 -> first kernel %+.3f | kernels+copies span %.3f | last end -> sync return %+.3f | t1 -> sync %.3f" % (up, first_k - t1, last - first_k, t1 + sync - last, sync))
    for e in mine:
        if e[0] < first_k + 0.03 or e[1] > last - 0.12: print("      %+8.3f .. %+8.3f  %s" % (e[0] - t1, e[1] - t1, e[2][:60]))
PY
find gpurun_out/gap_${tag} -name "*.csv" -size +1M -delete
