cd /tmp; export TMPDIR=/tmp; root=$GRAFT_REPO_ROOT
for v in runs runs29; do
ZK_MSM_HACC=$v ZK_MSM_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_one_$v -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/prof_one_$v.log 2>&1
st=$(find $root/gpurun_out/prof_one_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; grep -E "k_hacc|k_hsort|k_bitsum|k_ntt|k_wacc|k_wtail|k_wsort|k_r1cs" $st | awk -F, '{printf "%-60s calls %s avg %s ns\n", substr($1,1,60), $2, $4}'
find $root/gpurun_out/prof_one_$v -name "*.csv" -size +1M -delete
done
