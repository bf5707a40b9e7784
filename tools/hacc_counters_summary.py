#!/usr/bin/env python3
"""Summary of tools/hacc_counters.sh's passes: counters of k_hacc_runs29 per launch, the kernel's duration in every pass, the effective shader clock
(GRBM_GUI_ACTIVE, summed over the 8 XCDs, over the duration of the launches OF THE PASS THAT COUNTED IT: a profiled pass runs a few per cent slower than the plain
kernel trace) and the ratios that say where a wave's time goes.  python tools/hacc_counters_summary.py gpurun_out/hacc_<tag>"""
import csv, glob, json, collections, sys
out = sys.argv[1]; res = {"kernel": "k_hacc_runs29<0>", "mode": "ZK_MSM_ONE_STREAM=1 (alone on the chip), send circuit, 12 resident proofs per pass", "counters": {}, "duration_us": {}, "notes": []}
def durations(p):
    fs = glob.glob(out + "/" + p + "/**/*kernel_trace.csv", recursive=True)
    return sorted((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(fs[0])) if "k_hacc_runs29" in r["Kernel_Name"]) if fs else []
pass_of = {}
for p in ("trace", "pass1", "pass2", "pass3"):
    d = durations(p)
    if d: res["duration_us"][p] = {"median": d[len(d) // 2], "mean": sum(d) / len(d), "min": d[0], "max": d[-1], "launches": len(d)}
    fs = glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if "k_hacc_runs29" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    acc = collections.defaultdict(list)
    for d_ in per.values():
        for k, v in d_.items(): acc[k].append(v)
    for k, v in acc.items(): res["counters"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v), "pass": p}; pass_of[k] = p
c = {k: v["per_launch_mean"] for k, v in res["counters"].items()}; der = {}
wall = res["duration_us"].get("trace", {}).get("median"); res["wall_us_median_kernel_trace_only"] = wall; res["wall_us_mean_kernel_trace_only"] = wall   # (the median: a 12-launch mean carries any one preempted launch)
if "GRBM_GUI_ACTIVE" in c:
    dp = res["duration_us"][pass_of["GRBM_GUI_ACTIVE"]]["mean"]; clk = c["GRBM_GUI_ACTIVE"] / 8 / (dp * 1e3); der["effective_clock_GHz"] = round(clk, 3)
    der["effective_clock_note"] = "GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the launch duration in the same pass (%.1f us).  The kernel is power-limited: not the 2.4 GHz of the data sheet" % dp
    if "SQ_INSTS_VALU" in c and wall: der["valu_wave_instructions_per_launch"] = c["SQ_INSTS_VALU"]; der["cycles_per_valu_instruction_per_simd_at_measured_clock"] = round(wall * 1e-6 * clk * 1e9 * 1024 / c["SQ_INSTS_VALU"], 3)
if "SQ_WAVES" in c: der["waves"] = c["SQ_WAVES"]; der["rounds_of_4096_wave_slots"] = round(c["SQ_WAVES"] / 4096, 3)
if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c: der["mean_waves_per_simd_while_busy"] = round(4 * c["SQ_WAVE_CYCLES"] / ((c["SQ_BUSY_CYCLES"] / 32) * 1024), 3); der["mean_waves_note"] = "SQ_WAVE_CYCLES counts quad-cycles per wave, SQ_BUSY_CYCLES cycles per shader engine (32)"
for a in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    if a in c and "SQ_WAVE_CYCLES" in c: der[a + " / SQ_WAVE_CYCLES"] = round(c[a] / c["SQ_WAVE_CYCLES"], 4)
res["derived"] = der
print(json.dumps(res, indent=1))
