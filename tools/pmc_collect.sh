#!/bin/bash
# HBM traffic of the prover's kernels from the L2 memory-side counters: one rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md), kernel trace only.  Usage (on the GPU box, from the repo root):  bash tools/pmc_collect.sh <tag>
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$c -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $root
python3 tools/pmc_summarize.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE ${tag} > gpurun_out/pmc_${tag}_summary.json
find gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE -name "*.csv" -size +2M -delete
