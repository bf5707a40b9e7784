#!/bin/bash
# Is the step time of one build bimodal from process to process, and does keeping the scan threads beside the caller's buffer (hostnuma.hpp, ZK_NUMA_PIN) end that?
# Fresh processes, alternating, each 400 steps, with the prover's own clocks; then bench.py bound to the GPU's socket and unbound.
for rep in 1 2 3 4 5; do
  for pin in 1 0; do echo "ZK_NUMA_PIN=$pin: $(ZK_NUMA_PIN=$pin python tools/step_times.py 400 2>&1 | tail -2 | tr '\n' ' ')"; done
done
for bind in 1 0 1 0; do echo "bench ZK_BENCH_BIND=$bind: $(ZK_BENCH_BIND=$bind python bench.py --steps 100 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["step_ms"], d["prover_timings_ms"], d["config"]["host_binding"])')"; done
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocm-smi --showtoponuma 2>/dev/null | grep -i numa
