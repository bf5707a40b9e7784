#!/bin/bash
# Does the page size behind the caller's assignment explain the hand-over's spread from process to process?  Fresh processes: numpy's own allocation / huge pages advised / refused.
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
for rep in 1 2 3 4; do
  for mode in default huge small; do echo "STEP_PAGES=$mode: $(STEP_PAGES=$mode python tools/step_times.py 400 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-400)"; done
done
for ll in 1 2 1 2; do echo "ZK_MSM_COMBINE_LL=$ll: $(ZK_MSM_COMBINE_LL=$ll STEP_PAGES=huge python tools/step_times.py 400 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-400)"; done
