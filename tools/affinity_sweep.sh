#!/bin/bash
# compact CPU sets against the kernel's placement over all 256 hardware threads (fresh process each; host-buffer send proofs, tools/step_times.py)
rocm-smi --showtoponuma 2>/dev/null | grep -i "numa node"; lscpu | grep -i "numa node[01]"
for rep in 1 2; do
  for set in all 0-7 0-15 0-31 64-71 64-79 64-95 0-7,128-135 64-71,192-199; do
    if [ $set = all ]; then pre=""; else pre="taskset -c $set"; fi
    echo "cpus $set: $($pre python tools/step_times.py 300 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)"
  done
done
