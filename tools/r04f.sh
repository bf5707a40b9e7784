#!/bin/bash
# round 4, sixth GPU session: fixed-base table caps for the depth-32 deposit key, wave-priority variants against the default build on ONE box, the genSendproof path with the polling task pool
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
for cap in 768 1400 3000 6000; do echo "== ZK_MSM_PRECOMPUTE_MAX_MB=$cap"; ZK_CB_STAGES=1 ZK_MSM_PRECOMPUTE_MAX_MB=$cap timeout 900 python tools/circuit_bench.py deposit:32 2>&1 | tail -4 | cut -c1-1500; done > gpurun_out/r04f_deposit32_caps.txt 2>&1
cp blockmaze_amd/libzkgpu.so /tmp/default_lib.so
for rep in 1 2; do
  echo "default: $(python tools/step_times.py 600 2>&1 | tail -1)"
  for t in a b c; do cp tools/prio_$t.bin blockmaze_amd/libzkgpu.so; echo "prio_$t: $(python tools/step_times.py 600 2>&1 | tail -1)"; cp /tmp/default_lib.so blockmaze_amd/libzkgpu.so; done
done > gpurun_out/r04f_prio_ab.txt 2>&1
python tools/abi_step_times.py 300 > gpurun_out/r04f_abi_steps.txt 2>&1
python tools/abi_trace.py 2>&1 | grep trace-abi | tail -6 >> gpurun_out/r04f_abi_steps.txt
