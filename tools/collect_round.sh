python bench.py > gpurun_out/r03s_bench_default.json 2> gpurun_out/r03s_bench_default.err
bash tools/prof_collect.sh r03s
bash tools/pmc_insts.sh r03s
python tools/circuit_bench.py > gpurun_out/r03s_circuits.txt 2>&1
python tools/soak.py 2000 6 2>&1 | grep -v "Trying to generate" > gpurun_out/r03s_soak.txt
python tools/verify_bench.py > gpurun_out/r03s_verify_batch.txt 2>&1; python tools/verify_bench.py abi >> gpurun_out/r03s_verify_batch.txt 2>&1; ZK_VERIFY_GPU_MIN=1000000 python tools/verify_bench.py abi >> gpurun_out/r03s_verify_batch.txt 2>&1
bash tools/gap_probe.sh r03s
