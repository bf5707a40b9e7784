#!/bin/bash
# Kernel statistics + one-proof timeline of a bench run.  Usage (on the GPU box, from the repo root):  bash tools/prof_collect.sh <tag>
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag} -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/prof_${tag}.log 2>&1
cd $root
st=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof_${tag} -name "*kernel_trace.csv" | head -1)
cp "$st" gpurun_out/${tag}_kernel_stats.csv
python3 tools/timeline.py "$tr" 10 > gpurun_out/${tag}_timeline_one_proof.txt 2>&1
tail -1 gpurun_out/prof_${tag}.log > gpurun_out/${tag}_bench_line.json
find gpurun_out/prof_${tag} -name "*.csv" -size +2M -delete
