for rep in 1 2 3; do
  for st in 0000 0011 0010 0001; do
    echo "start=$st: $(ZK_WMSM_START=$st python tools/step_times.py 600 2>&1 | tail -1)"
  done
done
for st in 0000 0011; do echo "bench start=$st: $(ZK_WMSM_START=$st python bench.py --steps 100 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["step_ms"], d["prover_timings_ms"])')"; done
