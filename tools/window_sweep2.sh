#!/bin/bash
# window sizes of the H-query MSM and of the witness MSMs inside the whole proof (ms per proof; two rounds)
for rep in 1 2; do for hw in 15 16 17; do for ww in 7 8 9 10; do echo -n "H=$hw W=$ww  "; ZK_MSM_H_WINDOW=$hw ZK_MSM_WITNESS_WINDOW=$ww python bench.py --steps 60 --warmup 6 --no-cpu-baseline --inflight 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done; done
