#!/bin/bash
# A/B of two builds of libzkgpu.so on ONE box: tools/old_libzkgpu.bin (a previous build, copied there by hand) against the in-tree library.  bash tools/ab_lib.sh [ab_bench variants...]
cp blockmaze_amd/libzkgpu.so /tmp/new_lib.so
for rep in 1 2; do
  echo "--- new build"; python tools/ab_bench.py --rounds 2 "$@" 2>&1 | grep "==\|ntt"
  cp tools/old_libzkgpu.bin blockmaze_amd/libzkgpu.so; echo "--- old build"; python tools/ab_bench.py --rounds 2 "$@" 2>&1 | grep "==\|ntt"; cp /tmp/new_lib.so blockmaze_amd/libzkgpu.so
done
