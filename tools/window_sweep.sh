#!/bin/bash
# sweep of the window sizes of the H-query MSM (ZK_MSM_H_WINDOW) and of the witness MSMs (ZK_MSM_WITNESS_WINDOW)
for hw in 13 14 15 16; do for ww in ${WW:-8}; do echo "H=$hw W=$ww"; ZK_MSM_H_WINDOW=$hw ZK_MSM_WITNESS_WINDOW=$ww python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); s=d['stage_ms_per_proof']; print({k:v for k,v in s.items() if k.startswith('msm_H')})"; done; done
