#!/bin/bash
# The spread of the hand-over from process to process: eight fresh processes of the in-tree build, 400 steps each, with the prover's own clocks
for rep in 1 2 3 4 5 6 7 8; do python tools/step_times.py 400 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330; echo; done
