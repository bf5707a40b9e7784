#!/bin/bash
# run length of the H accumulation, finer than round 4's first sweep (12 / 16 / 20 / 24): whole proofs, fresh processes, the process confined to 32 cores like bench.py's rank
for rep in 1 2 3; do
  for r in 12 11 10; do echo "send ZK_MSM_H_RUN=$r: $(ZK_MSM_H_RUN=$r taskset -c 0-31 python tools/step_times.py 300 2>&1 | tail -1 | cut -c1-150)"; done
done
for rep in 1 2; do
  for r in 12 11 10 13; do echo "== ZK_MSM_H_RUN=$r"; ZK_CB_N=100 ZK_MSM_H_RUN=$r taskset -c 0-31 python tools/circuit_bench.py mint:8 redeem:8 deposit:8 deposit:32 2>&1 | grep "ms/proof" | cut -c1-20,150-200; done
done
