#!/bin/bash
# VALU instructions per kernel and per proof (SQ_INSTS_VALU summed over a bench run, divided by the number of H accumulations = proofs): where the proof's issue slots go.
# bash tools/pmc_insts.sh <tag>     (on the GPU box, from the repo root; counters in their own run, kernel trace only)
tag=${1:-r03}; root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
ZK_MSM_ONE_STREAM=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $root/gpurun_out/pmc_insts_${tag} -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/pmc_insts_${tag}.log 2>&1
cd $root
python3 - <<PY > gpurun_out/${tag}_valu_insts_per_proof.txt
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_insts_${tag}/**/*counter_collection.csv", recursive=True)[0]; acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", ""); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); calls[k] += 1
proofs = max(1, sum(c for k, c in calls.items() if k.startswith("k_hacc_runs29")))
tot = sum(c["SQ_INSTS_VALU"] for c in acc.values())
print("proofs in the run: %d; VALU wave-instructions per proof: %.4g (all kernels of the run, key load excluded only in so far as its kernels are named differently)" % (proofs, tot / proofs))
print("%-56s %8s %12s %7s %10s %10s %10s" % ("kernel", "calls/pf", "VALU/proof", "share", "SALU/pf", "LDS/pf", "VMEM_RD/pf"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    print("%-56s %8.2f %12.4g %6.1f%% %10.3g %10.3g %10.3g" % (k[:56], calls[k] / proofs, c["SQ_INSTS_VALU"] / proofs, 100 * c["SQ_INSTS_VALU"] / tot, c["SQ_INSTS_SALU"] / proofs, c["SQ_INSTS_LDS"] / proofs, c["SQ_INSTS_VMEM_RD"] / proofs))
PY
find gpurun_out/pmc_insts_${tag} -name "*.csv" -size +1M -delete
