#!/bin/bash
# per-step time distribution of two builds on one box (tools/old_libzkgpu.bin against the in-tree library)
cp blockmaze_amd/libzkgpu.so /tmp/new_lib.so
for rep in 1 2; do
  echo "new: $(python tools/step_times.py 800 2>&1 | tail -1)"
  cp tools/old_libzkgpu.bin blockmaze_amd/libzkgpu.so; echo "old: $(python tools/step_times.py 800 2>&1 | tail -1)"; cp /tmp/new_lib.so blockmaze_amd/libzkgpu.so
done
