#!/bin/bash
for sg in 8 16 32 64; do echo "SEG=$sg"; ZK_MSM_SEG=$sg python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); s=d['stage_ms_per_proof']; print({k:v for k,v in s.items() if k.startswith('msm_H')})"; done
