#!/bin/bash
# round 4, ninth GPU session: the whole GPU suite on the current build, the default bench line, profiles, hand-over trace
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04i_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04i_tests.log
python bench.py > gpurun_out/r04i_bench_default.json 2> gpurun_out/r04i_bench_default.err; echo "bench rc $?" >> gpurun_out/r04i_bench_default.err
bash tools/prof_collect.sh r04i
python tools/abi_trace.py 2>&1 | grep -E "trace-abi|trace-handover" | tail -12 > gpurun_out/r04i_abi_trace.txt
python tools/circuit_bench.py > gpurun_out/r04i_circuits.txt 2>&1
