#!/bin/bash
# round 4, fourth GPU session: half-row workgroups in the marginal kernel, the H query's window size re-swept with the new tail, the bench line with the new genSendproof leg
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests/test_gpu_engine.py tests/test_gpu_groth16.py -m gpu -x -q -k "msm or proof_bytes_match or send_proof_full_size" > gpurun_out/r04d_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04d_tests.log
for rep in 1 2; do for v in "" "ZK_MSM_H_WINDOW=15" "ZK_MSM_H_WINDOW=17"; do echo "[$v] $(env $v python tools/step_times.py 400 2>&1 | tail -1)"; done; done > gpurun_out/r04d_ab.txt 2>&1
python bench.py > gpurun_out/r04d_bench_default.json 2> gpurun_out/r04d_bench_default.err; echo "bench rc $?" >> gpurun_out/r04d_bench_default.err
bash tools/prof_collect.sh r04d
