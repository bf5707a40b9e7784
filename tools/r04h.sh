#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout 1200 python tools/dbg_g2.py > gpurun_out/r04h_dbg_g2.txt 2>&1
python tools/abi_step_times.py 300 > gpurun_out/r04h_abi_steps.txt 2>&1
python tools/abi_trace.py 2>&1 | grep trace-abi | tail -6 >> gpurun_out/r04h_abi_steps.txt
