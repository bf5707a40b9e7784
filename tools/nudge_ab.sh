#!/bin/bash
# the wake-up of the scan pool at the end of a proof's device work (ZK_SCAN_NUDGE), interleaved fresh processes; bench.py's 55 distinct assignments
for rep in 1 2 3 4; do for v in 1 0; do echo "ZK_SCAN_NUDGE=$v: $(ZK_SCAN_NUDGE=$v python tools/handover_trace.py 400 55) $(ZK_SCAN_NUDGE=$v python tools/step_times.py 300 2>&1 | tail -1 | cut -c1-140)"; done; done
