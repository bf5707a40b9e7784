#!/bin/bash
# round 4, second GPU session: tagged rows kernel / tagged witness sort / one workgroup per weight bit in the witness tails — parity, A/B, timeline, genSendproof breakdown
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_full.py tests/test_gpu_groth16.py -m gpu -x -q -k "not libsnark_generator and not depth32_single and not c_driver and not key_container and not key_generation and not verifier" > gpurun_out/r04b_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04b_tests.log
for v in "" "ZK_ROWS_TAGGED=0" "ZK_WSORT_TAGGED=0" "ZK_ROWS_TAGGED=0 ZK_WSORT_TAGGED=0"; do echo "[$v] $(env $v python tools/step_times.py 400 2>&1 | tail -1)"; done > gpurun_out/r04b_ab.txt 2>&1
for v in "" "ZK_ROWS_TAGGED=0 ZK_WSORT_TAGGED=0"; do echo "[$v] $(env $v python tools/step_times.py 400 2>&1 | tail -1)"; done >> gpurun_out/r04b_ab.txt 2>&1
bash tools/prof_collect.sh r04b
python tools/abi_trace.py > gpurun_out/r04b_abi_trace.txt 2>&1
python tools/abi_step_times.py 300 > gpurun_out/r04b_abi_steps.txt 2>&1
python tools/circuit_bench.py > gpurun_out/r04b_circuits.txt 2>&1
