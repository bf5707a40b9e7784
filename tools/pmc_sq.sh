#!/bin/bash
# Where the waves of the H accumulation spend their cycles: SQ counters of one pass (8 SQ slots), everything on one stream.  bash tools/pmc_sq.sh <tag> [ZK_MSM_HACC value]
tag=${1:-r03}; v=${2:-runs29}; root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
ZK_MSM_HACC=$v ZK_MSM_ONE_STREAM=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $root/gpurun_out/pmc_sq_${tag}_$v -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/pmc_sq_${tag}_$v.log 2>&1
cd $root
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_sq_${tag}_$v/**/*counter_collection.csv", recursive=True)[0]; acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)): acc[r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if not any(x in k for x in ("k_hacc", "k_bitsum_chunks", "k_ntt_cols", "k_wacc_lanes")): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}; wc = m.get("SQ_WAVE_CYCLES", 1)
    print("%-44s" % k[:44], " ".join("%s=%.3g" % (n.replace("SQ_", ""), x) for n, x in sorted(m.items())), "| wait_any/wave_cycles %.2f  wait_inst/wave_cycles %.2f  active_valu/wave_cycles %.2f" % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_VALU", 0) / wc))
PY
find gpurun_out/pmc_sq_${tag}_$v -name "*.csv" -size +1M -delete
