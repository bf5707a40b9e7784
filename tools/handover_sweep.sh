#!/bin/bash
# hand-over breakdown against the number of scan threads and the helper threads' polling time (fresh process each; 55 distinct assignments as in bench.py)
for rep in 1 2; do
  for t in 16 12 8 4; do echo "ZK_SCAN_THREADS=$t: $(ZK_SCAN_THREADS=$t python tools/handover_trace.py 400 55)"; done
  for us in 1200; do echo "ZK_SPIN_US=$us (16 threads): $(ZK_SPIN_US=$us python tools/handover_trace.py 400 55)"; done
done
