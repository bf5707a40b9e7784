#!/bin/bash
# round 4, fifth GPU session: round-3 library against the current one on ONE box, fixed-base table caps for the depth-32 deposit key, where a genSendproof call's time goes inside bench.py
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
bash tools/ab_steps.sh > gpurun_out/r04e_ab_old_new.txt 2>&1
for cap in 768 1400 3000 6000; do echo "== ZK_MSM_PRECOMPUTE_MAX_MB=$cap"; ZK_CB_STAGES=1 ZK_MSM_PRECOMPUTE_MAX_MB=$cap timeout 600 python tools/circuit_bench.py deposit:32 2>&1 | tail -3; done > gpurun_out/r04e_deposit32_caps.txt 2>&1
ZK_TRACE_TIMES=1 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r04e_bench_trace.json 2> gpurun_out/r04e_bench_trace.err
grep -E "trace-abi" gpurun_out/r04e_bench_trace.err | head -80 > gpurun_out/r04e_abi_in_bench.txt
python tools/abi_step_times.py 300 > gpurun_out/r04e_abi_steps.txt 2>&1
