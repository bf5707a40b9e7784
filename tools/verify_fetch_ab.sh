#!/bin/bash
# A/B of the ways kernel K9 fetches its instruction words (ZK_VERIFY_FETCH: 0 plain load per round, 1 ring of registers, 2 ring in LDS): verdicts of the K9 tests, then device times
for f in 0 1 2; do
  echo "== ZK_VERIFY_FETCH=$f"
  ZK_VERIFY_FETCH=$f python -m pytest tests/test_gpu_groth16.py -x -q -m gpu -k "batched_gpu_verifier_matches_host_verifier" 2>&1 | tail -2
  ZK_VERIFY_FETCH=$f python tools/verify_bench.py 2>&1 | grep -E "n = +(1|64|512):|Error|error" | cut -c1-200
done
