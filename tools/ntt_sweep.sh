#!/bin/bash
for r in 1 2 3; do for c in 0 1; do echo "radix_log=$r logC=$c"; ZK_NTT_RADIX_LOG=$r ZK_NTT_LOGC=$c python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); s=d['stage_ms_per_proof']; print({k:v for k,v in s.items() if k.startswith('ntt')})"; done; done
