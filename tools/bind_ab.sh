for rep in 1 2 3; do for b in 1 0; do echo "ZK_BENCH_BIND=$b: $(ZK_BENCH_BIND=$b python bench.py --steps 200 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["step_ms"], d["config"]["host_binding"])')"; done; done
python -m pytest tests/test_gpu_multi_rank.py -m gpu -x -q 2>&1 | tail -2
