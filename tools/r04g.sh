#!/bin/bash
# round 4, seventh GPU session: the G2 witness MSM lane by lane on 29-bit limbs (parity, A/B, timeline), fixed-base table caps for the depth-32 deposit key, genSendproof steps
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_groth16.py tests/test_gpu_parity_full.py -m gpu -x -q -k "msm or proof_bytes_match or send_proof_full_size or g2 or witness_msm or switch or cut_into" > gpurun_out/r04g_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04g_tests.log
for rep in 1 2; do for v in "" "ZK_G2_LANES29=0"; do echo "[$v] $(env $v python tools/step_times.py 600 2>&1 | tail -1)"; done; done > gpurun_out/r04g_ab.txt 2>&1
bash tools/prof_collect.sh r04g
for cap in 768 1400 3000 6000; do echo "== ZK_MSM_PRECOMPUTE_MAX_MB=$cap"; ZK_CB_STAGES=1 ZK_MSM_PRECOMPUTE_MAX_MB=$cap timeout 900 python tools/circuit_bench.py deposit:32 2>&1 | tail -2 | cut -c1-1800; done > gpurun_out/r04g_deposit32_caps.txt 2>&1
python tools/abi_step_times.py 300 > gpurun_out/r04g_abi_steps.txt 2>&1
