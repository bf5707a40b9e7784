#!/bin/bash
# SQ / GRBM counters of the dominant kernel (k_hacc_runs29) ALONE on the chip (ZK_MSM_ONE_STREAM=1: everything on one stream), one rocprofv3 pass per counter set
# (8 SQ slots and 2 GRBM slots a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters in their own runs, kernel trace only), plus an un-profiled-counter kernel trace
# of the same command for the wall time.  bash tools/hacc_counters.sh <tag>   (on the GPU box, from the repo root) -> gpurun_out/<tag>_hacc_counters.json
tag=${1:-r05}; root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/hacc_${tag}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $out/counters_available.txt 2>&1
export ZK_MSM_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/trace_run.py send > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out/pass1 -- python3 $root/tools/trace_run.py send > $out/pass1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/pass2 -- python3 $root/tools/trace_run.py send > $out/pass2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_THREAD_CYCLES_VALU SQ_WAVES_LT_64 SQ_INSTS_BRANCH SQ_IFETCH SQ_CYCLES --kernel-trace --output-format csv -d $out/pass3 -- python3 $root/tools/trace_run.py send > $out/pass3.log 2>&1
cd $root
python3 - <<PY > gpurun_out/${tag}_hacc_counters.json
import csv, glob, json, collections
out = "gpurun_out/hacc_${tag}"; res = {"kernel": "k_hacc_runs29<0>", "mode": "ZK_MSM_ONE_STREAM=1 (alone on the chip), send circuit, 12 resident proofs per pass", "counters": {}, "notes": []}
for p in ("pass1", "pass2", "pass3"):
    fs = glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True)
    if not fs: res["notes"].append(p + ": no counter file (see " + p + ".log)"); continue
    acc = collections.defaultdict(list); per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "k_hacc_runs29" not in r["Kernel_Name"]: continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items(): acc[k].append(v)
    for k, v in acc.items(): res["counters"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
ts = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
if ts:
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(ts[0])) if "k_hacc_runs29" in r["Kernel_Name"]]
    if d: res["wall_us_mean_kernel_trace_only"] = sum(d) / len(d) / 1e3; res["wall_us_min"] = min(d) / 1e3; res["launches_timed"] = len(d)
c = {k: v["per_launch_mean"] for k, v in res["counters"].items()}; w = res.get("wall_us_mean_kernel_trace_only")
der = {}
if "GRBM_GUI_ACTIVE" in c and w: der["effective_clock_GHz (GRBM_GUI_ACTIVE / wall; the counter is summed over the XCDs if > 3: then / 8)"] = c["GRBM_GUI_ACTIVE"] / (w * 1e3)
if "SQ_INSTS_VALU" in c and w: der["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]; der["cycles_per_valu_instruction_per_simd_at_2.4GHz"] = w * 1e-6 * 2.4e9 * 1024 / c["SQ_INSTS_VALU"]
for a in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    if a in c and "SQ_WAVE_CYCLES" in c: der[a + " / SQ_WAVE_CYCLES"] = c[a] / c["SQ_WAVE_CYCLES"]
if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c: der["mean_waves_in_flight_per_SQ (WAVE_CYCLES / BUSY_CYCLES)"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
res["derived"] = der
print(json.dumps(res, indent=1))
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete 2>/dev/null
tail -c 1500 gpurun_out/${tag}_hacc_counters.json
