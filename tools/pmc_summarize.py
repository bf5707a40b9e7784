#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_summarize.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run>  > summary.json

Corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads, so it is doubled for STREAMING kernels — and only for those: calibrated on known byte counts (tools/gather_probe.hip, profiles/r03_pmc_calibration.json)
the factor is 2.00 for a coalesced 16-byte-per-lane read, 0.98 for random gathers of 64-byte records (the pattern of the bucket accumulations: their requests are
64 bytes and are counted as such) and 1.35 for 128-byte records.  Every kernel below carries its raw value, the factor applied and the pattern it was classed as;
WRITE_SIZE is taken as is.  Both are checked in the same run on k_qap_pointwise, a pure streaming kernel with a known
byte count (reads 3 vectors and writes 1 vector of m field elements of 32 bytes) or, in builds where that step is fused into the MSM sort, on k_fr_to_mont (the in-place
conversion of the assignment: n x 32 bytes read and written): the `calibration` entry holds measured/expected for both.
"""
import csv, glob, json, os, sys, collections
def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]; acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fetch = load(sys.argv[1], "FETCH_SIZE"); write = load(sys.argv[2], "WRITE_SIZE")
GATHER64 = ("k_hacc_runs29", "k_msm_accumulate_tasks<Fq>", "k_msm_accumulate_tasks<Fp", "k_wacc_lanes<Fq>", "k_wacc_lanes<Fp", "k_wacc_lanes29", "k_msm_sum_ones<Fq>")   # 64-byte point records fetched by index
GATHER128 = ("k_msm_accumulate_tasks<Fq2>", "k_wacc_quads<Fq2>", "k_wacc_lanes_g2_29", "k_msm_sum_ones<Fq2>")
try: CAL = {k.strip(): v["factor_expected_over_raw"] for k, v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03_pmc_calibration.json")))["patterns"].items()}
except Exception: CAL = {}
F_STREAM, F_G64, F_G128 = CAL.get("k_stream", 2.0), CAL.get("void k_gather<64>", 1.0), CAL.get("void k_gather<128>", 1.35)
def pattern(name):
    name = name.replace(" ", "")
    return ("gather64", F_G64) if name.startswith(GATHER64) else ("gather128", F_G128) if name.startswith(GATHER128) else ("stream", F_STREAM)
def short(n): return n.replace("zk::", "").replace("Fp<FqParams>", "Fq").replace("Fp<FrParams>", "Fr").split("(")[0].replace("void ", "")
out = {"tag": sys.argv[3] if len(sys.argv) > 3 else "untagged", "units": "bytes per launch; hbm_bytes = factor * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024, factor by access pattern (profiles/r03_pmc_calibration.json)", "kernels": {}}
for k in sorted(set(fetch) | set(write), key=lambda k: (short(k[0]), k[1])):
    f = fetch.get(k, (0.0, 0)); w = write.get(k, (0.0, 0))
    pat, fac = pattern(short(k[0]))
    out["kernels"]["%s grid=%d" % (short(k[0]), k[1])] = {"launches": max(f[1], w[1]), "FETCH_SIZE_KB_raw": round(f[0], 2), "WRITE_SIZE_KB": round(w[0], 2), "pattern": pat, "fetch_factor": fac, "hbm_bytes_per_launch": int(fac * f[0] * 1024 + w[0] * 1024)}
# the dominant kernel: bucket accumulation of the H-query MSM = the k_msm_accumulate_tasks<Fq> launch that moves the most bytes (4.2 M point gathers; the witness MSMs have a few 10^4)
acc = [(k, v) for k, v in out["kernels"].items() if k.startswith("k_hacc_runs29")]
if acc:
    name, v = max(acc, key=lambda kv: kv[1]["hbm_bytes_per_launch"]); out["k_msm_accumulate_H"] = dict(v, kernel=name)
pw = [(k, v) for k, v in out["kernels"].items() if k.startswith("k_qap_pointwise")]
fm = [(k, v) for k, v in out["kernels"].items() if k.startswith("k_fr_to_mont")]
if pw or fm:
    if pw: name, v = pw[0]; m = int(name.split("grid=")[1]); exp_r, exp_w = 3 * m * 32, m * 32       # reads a, b, c, writes a
    else: name, v = max(fm, key=lambda kv: int(kv[0].split("grid=")[1])); m = int(name.split("grid=")[1]); exp_r, exp_w = m * 32, m * 32   # in-place conversion of the assignment (grid = n rounded up to 256)
    out["calibration"] = {"kernel": name, "expected_read_bytes": exp_r, "expected_write_bytes": exp_w, "corrected_read_over_expected": round(F_STREAM * v["FETCH_SIZE_KB_raw"] * 1024 / exp_r, 3), "write_over_expected": round(v["WRITE_SIZE_KB"] * 1024 / exp_w, 3)}
print(json.dumps(out, indent=1))
