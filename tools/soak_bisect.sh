#!/bin/bash
# the mixed soak under the switches that select this round's new paths: which one loses proofs?
for cfg in "X=1" "ZK_HANDOVER_MAPPED=0" "ZK_MERGE_EQUAL_COLUMNS=0" "ZK_WITNESS_THREADS=0" "ZK_VERIFY_GPU_MIN=1000000"; do
  echo "== $cfg"; env $cfg python tools/soak_mixed.py ${1:-300} 4 2>&1 | grep -E "rejected|can not generate|repeated|libzkgpu" | cut -c1-260 | head -8
done
