#!/bin/bash
# Where the witness MSMs are released relative to the main chain (ZK_WMSM_START: one digit per job B2, L, A[+L], B1[+B2]; 0 = at once, 1 = after the row kernel, 2..4 = after the
# transforms): median step time of a host-buffer send proof for each setting, interleaved over two rounds.  bash tools/release_sweep.sh > gpurun_out/release_sweep.txt
for rep in 1 2; do
  for st in 0000 0011 0044 0040 0004 0014 0041; do
    echo "start=$st: $(ZK_WMSM_START=$st python tools/step_times.py 600 2>&1 | tail -1)"
  done
done
