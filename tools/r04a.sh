#!/bin/bash
# round 4, first GPU session: the new parity legs, the default bench line, kernel stats + timelines, and a sweep of the H accumulation's run length
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "libsnark_generator or full_size_against_libsnark or depth32_single or h_path_degenerate or verifier_failure or multi_rank" > gpurun_out/r04a_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04a_tests.log
python bench.py > gpurun_out/r04a_bench_default.json 2> gpurun_out/r04a_bench_default.err; echo "bench rc $?" >> gpurun_out/r04a_bench_default.err
bash tools/prof_collect.sh r04a
for r in 12 16 20 24; do echo "h_run $r: $(ZK_MSM_H_RUN=$r python tools/step_times.py 400 2>&1 | tail -1)"; done > gpurun_out/r04a_hrun_sweep.txt 2>&1
