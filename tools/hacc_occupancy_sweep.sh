#!/bin/bash
# k_hacc_runs29 at 4 / 3 / 2 waves per SIMD (unused dynamic LDS caps the workgroups per CU) x run lengths: kernel time alone on the chip (one stream, HIP-event stage clock of the prover)
# and whole proofs.  bash tools/hacc_occupancy_sweep.sh   (GPU box, repo root)
for lds in 0 41 54; do for run in 9 10 11 12 13 15 16; do
  echo "ZK_HACC_DYNLDS_KB=$lds ZK_MSM_H_RUN=$run: $(ZK_HACC_DYNLDS_KB=$lds ZK_MSM_H_RUN=$run python tools/step_times.py 200 2>&1 | tail -2 | head -1 | cut -c1-140) | one stream: $(ZK_MSM_ONE_STREAM=1 ZK_HACC_DYNLDS_KB=$lds ZK_MSM_H_RUN=$run python tools/stage_times.py msm_H 2>&1 | tail -1)"
done; done
