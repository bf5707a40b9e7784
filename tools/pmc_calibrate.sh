#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/gather_probe.hip): bash tools/pmc_calibrate.sh <tag>   ->  gpurun_out/<tag>_pmc_calibration.json
tag=${1:-r03}; root=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
[ -x $root/tools/gather_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $root/tools/gather_probe.bin $root/tools/gather_probe.hip
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_cal_${tag} -- $root/tools/gather_probe.bin > $root/gpurun_out/pmc_cal_${tag}.log 2>&1
cd $root; python3 - <<PY
import csv, glob, json, collections
f = glob.glob("gpurun_out/pmc_cal_${tag}/**/*counter_collection.csv", recursive=True)[0]; acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
lanes, per, table = 349504, 12, 16 * 262144 * 64; exp = {"k_stream": table, "void k_gather<64>": lanes * per * 64, "void k_gather<128>": lanes * per * 128}; out = {"tag": "${tag}", "note": "FETCH_SIZE is reported in KB; factor = expected bytes / (raw KB x 1024); MI355X_MICROARCH.md documents 2.0 for wide coalesced reads", "patterns": {}}
for k, v in acc.items():
    raw = sum(v) / len(v) * 1024; e = [x for n, x in exp.items() if k.strip().startswith(n)]
    if e: out["patterns"][k.strip()] = {"launches": len(v), "FETCH_SIZE_bytes_raw": int(raw), "expected_bytes": e[0], "factor_expected_over_raw": round(e[0] / raw, 3)}
json.dump(out, open("gpurun_out/${tag}_pmc_calibration.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
