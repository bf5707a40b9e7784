#!/bin/bash
# Does the page size behind the caller's assignment explain the hand-over's spread from process to process?  Fresh processes: numpy's own allocation / huge pages advised / refused.
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
for rep in 1 2 3 4; do
  for mode in default huge small; do echo "STEP_PAGES=$mode: $(STEP_PAGES=$mode python tools/step_times.py 400 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-400)"; done
done
