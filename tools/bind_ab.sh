#!/bin/bash
# bench.py confined to a quiet block of 32 neighbouring cores (the default) against the kernel's placement, interleaved; then what the placement sees of the host
for rep in 1 2 3; do for b in 1 0; do echo "ZK_BENCH_BIND=$b: $(ZK_BENCH_BIND=$b python bench.py --steps 200 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["step_ms"], d["config"]["host_binding"])')"; done; done
python -m pytest tests/test_gpu_multi_rank.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY'
import os, sys; sys.path.insert(0, os.getcwd())
from blockmaze_amd import sharding as placement
b = placement.cpu_busy_fractions(0.2); print("cpus sampled", len(b), "busy > 0.3:", sorted(c for c, v in b.items() if v > 0.3)[:40], "siblings of 0:", placement.cpu_siblings().get(0))
PY
