#!/bin/bash
# Which witness chain costs the transforms their time?  A diagnostic build (make HOOKS=1, copied to tools/hooks_lib.bin by hand) drops the A / L* pair ('a'), the B pair ('b')
# or all witness MSMs ('w') from every proof — the proofs are WRONG then, only the kernel trace means something.  bash tools/skip_probe.sh <tag>
# The diagnostic library must never stay where the shipped one is loaded from: the original comes back on EVERY way out of this script (trap), and nothing is
# overwritten before the hooks build is known to be there.
set -e
tag=${1:-r04}; root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
[ -s tools/hooks_lib.bin ] || { echo "tools/hooks_lib.bin (make HOOKS=1 build) is missing" >&2; exit 2; }
cp blockmaze_amd/libzkgpu.so /tmp/ship_lib.so; trap 'cp /tmp/ship_lib.so "$root/blockmaze_amd/libzkgpu.so"' EXIT INT TERM
cp tools/hooks_lib.bin blockmaze_amd/libzkgpu.so
set +e
cd /tmp; export TMPDIR=/tmp
for v in none a b w; do
  ZK_DEBUG_SKIP=$v rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/skip_${tag}_$v -- python3 $root/tools/trace_run.py send > /dev/null 2> $root/gpurun_out/skip_${tag}_$v.err
  tr=$(find $root/gpurun_out/skip_${tag}_$v -name "*kernel_trace.csv" | head -1)
  echo "== ZK_DEBUG_SKIP=$v"; python3 $root/tools/timeline.py "$tr" 6 | sed -n 1,16p
done > $root/gpurun_out/${tag}_skip_probe.txt 2>&1
find $root/gpurun_out/skip_${tag}_* -name "*.csv" -size +1M -delete
