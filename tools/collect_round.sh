#!/bin/bash
# Everything a round's profiles/ directory is made of, in one GPU session:  bash tools/collect_round.sh <tag>   (on the GPU box, from the repo root; ~12 minutes)
tag=${1:-r04s}
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
bash tools/prof_collect.sh ${tag}
grep -h "general MSM path" gpurun_out/prof_${tag}.err | sort | uniq -c > gpurun_out/${tag}_fallbacks.txt
bash tools/pmc_collect.sh ${tag}
bash tools/pmc_insts.sh ${tag}
# (the per-circuit figures and the second step-time run: the process confined to 32 neighbouring cores, as bench.py confines its rank)
ZK_CB_N=100 taskset -c 0-31 python tools/circuit_bench.py > gpurun_out/${tag}_circuits.txt 2>&1
python tools/step_times.py 600 > gpurun_out/${tag}_step_times.txt 2>&1
taskset -c 0-31 python tools/step_times.py 600 >> gpurun_out/${tag}_step_times.txt 2>&1
taskset -c 0-31 python tools/trace_tail.py 300 >> gpurun_out/${tag}_step_times.txt 2>&1
python tools/abi_step_times.py 400 > gpurun_out/${tag}_abi_step_times.txt 2>&1
python tools/abi_trace.py 2>&1 | grep -E "trace-abi|trace-handover" | tail -8 >> gpurun_out/${tag}_abi_step_times.txt
python tools/soak.py 2000 6 2>&1 | grep -v "Trying to generate" > gpurun_out/${tag}_soak.txt
python tools/verify_bench.py > gpurun_out/${tag}_verify_batch.txt 2>&1; python tools/verify_bench.py abi >> gpurun_out/${tag}_verify_batch.txt 2>&1
bash tools/gap_probe.sh ${tag}
