#!/bin/bash
# A/B of two builds of libzkgpu.so on ONE box by the default bench line: tools/old_libzkgpu.bin (a previous build, put there by hand) against the in-tree library, alternating.  bash tools/ab_lib_value.sh [reps]
reps=${1:-3}; cp blockmaze_amd/libzkgpu.so /tmp/new_lib.so
line() { python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); st=j['stage_ms_per_proof']
print('$1: value %.1f /s, p50 %.4f ms | fwd %.4f inv %.4f rows %.4f Hsort %.4f Hacc %.4f Hcomb %.4f Hred %.4f | host buffers %.1f /s' % (j['value'], j['step_ms']['p50'], st['ntt.forward'], st['ntt.inverse'], st['r1cs.rows'], st['msm_H.sort'], st['msm_H.accumulate'], st['msm_H.combine'], st['msm_H.reduce'], j['value_from_host_buffers']))"; }
for rep in $(seq $reps); do
  line new; cp tools/old_libzkgpu.bin blockmaze_amd/libzkgpu.so; line old; cp /tmp/new_lib.so blockmaze_amd/libzkgpu.so
done
