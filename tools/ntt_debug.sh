#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for d in 0 256 512 768; do export ZK_NTT_DEBUG=$d; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ntt_dbg$d -- python3 $GRAFT_REPO_ROOT/tools/ntt_probe.py > /dev/null 2>&1; f=$(find $GRAFT_REPO_ROOT/gpurun_out/ntt_dbg$d -name "*kernel_stats.csv" | head -1); echo "debug=$d"; grep -E "k_ntt_(cols|rows)" $f | sed -E 's/^"zk::(k_ntt_[a-z]*)[^"]*",([0-9]*),([0-9]*),([0-9.]*),.*/\1 calls \2 avg_ns \4/'; rm -rf $GRAFT_REPO_ROOT/gpurun_out/ntt_dbg$d; done
