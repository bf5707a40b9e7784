#!/bin/bash
# A/B of one environment switch on the default bench line (20 steps; device-resident statements): tools/ab_env.sh VAR v1 v2 ... — prints value, p50 step and the stage times of the roofline leg
var=$1; shift
for v in "$@"; do
  for rep in 1 2; do
    env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); st=j['stage_ms_per_proof']
print('$var=$v: value %.1f /s, ms_per_step %.4f, p50 %.4f | ntt fwd %.4f inv %.4f rows %.4f Hsort %.4f Hacc %.4f Hcomb %.4f Hred %.4f | host buffers %.1f /s' % (j['value'], j['ms_per_step'], j['step_ms']['p50'], st['ntt.forward'], st['ntt.inverse'], st['r1cs.rows'], st['msm_H.sort'], st['msm_H.accumulate'], st['msm_H.combine'], st['msm_H.reduce'], j['value_from_host_buffers']))"
  done
done
