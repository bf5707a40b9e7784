#!/bin/bash
# repeated comparison of the transform's radix / tile settings inside the whole proof (ms per proof; three runs each)
for cfg in "3 1" "2 0" "2 1" "2 2" "3 1" "2 0" "2 1" "2 2" "3 1" "2 0" "2 1" "2 2"; do set -- $cfg; echo -n "radix_log=$1 logC=$2  "; ZK_NTT_RADIX_LOG=$1 ZK_NTT_LOGC=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
