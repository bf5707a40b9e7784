#!/bin/bash
# SQ / GRBM counters of the dominant kernel (k_hacc_runs29) ALONE on the chip (ZK_MSM_ONE_STREAM=1: everything on one stream), one rocprofv3 pass per counter set
# (8 SQ slots and 2 GRBM slots a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters in their own runs, kernel trace only), plus an un-profiled-counter kernel trace
# of the same command for the wall time.  bash tools/hacc_counters.sh <tag>   (on the GPU box, from the repo root) -> gpurun_out/<tag>_hacc_counters.json
tag=${1:-r05}; root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/hacc_${tag}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $out/counters_available.txt 2>&1
export ZK_MSM_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/trace_run.py send > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/pass1 -- python3 $root/tools/trace_run.py send > $out/pass1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/pass2 -- python3 $root/tools/trace_run.py send > $out/pass2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_THREAD_CYCLES_VALU SQ_WAVES_LT_64 SQ_INSTS_BRANCH SQ_IFETCH SQ_CYCLES --kernel-trace --output-format csv -d $out/pass3 -- python3 $root/tools/trace_run.py send > $out/pass3.log 2>&1
cd $root
python3 tools/hacc_counters_summary.py gpurun_out/hacc_${tag} > gpurun_out/${tag}_hacc_counters.json
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete 2>/dev/null
tail -c 1500 gpurun_out/${tag}_hacc_counters.json
