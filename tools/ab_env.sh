#!/bin/bash
# A/B of one environment switch on one box: bash tools/ab_env.sh VAR valueA valueB [rounds] — interleaved fresh processes, tools/step_times.py (host-buffer send proofs) each
VAR=$1; A=$2; B=$3; R=${4:-3}
for rep in $(seq $R); do
  for v in "$A" "$B"; do echo "$VAR=$v: $(env $VAR=$v python tools/step_times.py 500 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)"; done
done
