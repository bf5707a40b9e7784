#!/bin/bash
# Kernel statistics + one-proof timeline of a bench run, once with the prover's five streams and once with everything on one stream (stand-alone kernel durations).
# Usage (on the GPU box, from the repo root):  bash tools/prof_collect.sh <tag>
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag} -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/prof_${tag}.log 2> $root/gpurun_out/prof_${tag}.err
ZK_MSM_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_1s -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/prof_${tag}_1s.log 2> $root/gpurun_out/prof_${tag}_1s.err
cd $root
for v in "" _1s; do
  st=$(find gpurun_out/prof_${tag}${v} -name "*kernel_stats.csv" | head -1); tr=$(find gpurun_out/prof_${tag}${v} -name "*kernel_trace.csv" | head -1)
  [ "$v" = "" ] && cp "$st" gpurun_out/${tag}_kernel_stats.csv || cp "$st" gpurun_out/${tag}_kernel_stats_one_stream.csv
  [ "$v" = "" ] && python3 tools/timeline.py "$tr" 10 > gpurun_out/${tag}_timeline_one_proof.txt 2>&1 || python3 tools/timeline.py "$tr" 10 > gpurun_out/${tag}_timeline_one_stream.txt 2>&1
done
grep '^{"metric"' gpurun_out/prof_${tag}.log | tail -1 > gpurun_out/${tag}_bench_line.json     # (the JSON line only: rocprofv3's own chatter goes to stderr / other lines)
find gpurun_out/prof_${tag} gpurun_out/prof_${tag}_1s -name "*.csv" -size +2M -delete
