#!/bin/bash
# round 4, third GPU session: the H tail on 29-bit limbs (marginal sums + bit sums), batched loads in the tagged rows kernel, one atomic per workgroup in the witness sorts
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity_full.py tests/test_gpu_groth16.py -m gpu -x -q -k "not libsnark_generator and not depth32_single and not c_driver and not key_container and not key_generation and not verifier and not full_size_against" > gpurun_out/r04c_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04c_tests.log
for rep in 1 2; do for v in "" "ZK_MSM_H_TAIL29=0" "ZK_ROWS_TAGGED=0" "ZK_WSORT_TAGGED=0"; do echo "[$v] $(env $v python tools/step_times.py 400 2>&1 | tail -1)"; done; done > gpurun_out/r04c_ab.txt 2>&1
bash tools/prof_collect.sh r04c
