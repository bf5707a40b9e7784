#!/bin/bash
# per-stream timeline of one proof of every circuit (tools/trace_run.py under rocprofv3 --kernel-trace):  bash tools/circuit_timelines.sh <tag>
tag=${1:-r04}; root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for c in "mint 8" "redeem 8" "deposit 8" "deposit 32"; do
  set -- $c; name=$1_$2
  rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tl_${tag}_$name -- python3 $root/tools/trace_run.py $1 $2 > /dev/null 2> $root/gpurun_out/tl_${tag}_$name.err
  tr=$(find $root/gpurun_out/tl_${tag}_$name -name "*kernel_trace.csv" | head -1)
  python3 $root/tools/timeline.py "$tr" 6 > $root/gpurun_out/${tag}_timeline_$name.txt 2>&1
done
find $root/gpurun_out/tl_${tag}_* -name "*.csv" -size +1M -delete
