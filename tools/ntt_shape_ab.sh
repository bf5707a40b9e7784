#!/bin/bash
# transform tile shape and radix after the round's instruction cuts: device_ms of a send proof (tools/step_times.py), fresh process each
for rep in 1 2; do
  for cfg in "1 2" "1 3" "0 2" "0 3" "2 2" "2 3"; do set -- $cfg
    echo "ZK_NTT_LOGC=$1 ZK_NTT_RADIX_LOG=$2: $(ZK_NTT_LOGC=$1 ZK_NTT_RADIX_LOG=$2 python tools/step_times.py 300 2>&1 | tail -1 | cut -c1-150)"
  done
done
