"""Summarise the rocprofv3 PMC passes of tools/pmc_dominant.sh (gpurun_out/pmcd_*) into profiles/<name>.json.
FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section); units are KB."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "pmc_dominant.json")
counters, durs, kname = {}, [], None
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmcd_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        acc, cnt = {}, {}
        for r in csv.DictReader(open(f)):
            if "conv_igemm_kernel" not in r["Kernel_Name"]:
                continue
            kname = r["Kernel_Name"]
            c = r["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
            cnt[c] = cnt.get(c, 0) + 1
        for c in acc:
            counters[c] = acc[c] / cnt[c]
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "conv_igemm_kernel" in r["Kernel_Name"]:
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
alg = (16 * 192 * 128 * 128 + 16 * 192 * 64 * 64 + 192 * 192 * 25) * 4
res = {"kernel": kname, "workload": "g_a.2 forward conv5x5 s2 192->192 on [16,192,128,128]",
       "avg_duration_us_under_pmc": sum(durs) / max(len(durs), 1), "counters": counters,
       "algorithmic_bytes_per_launch": alg}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    res["fetch_bytes_corrected_x2"] = counters["FETCH_SIZE"] * 1024 * 2
    res["write_bytes"] = counters["WRITE_SIZE"] * 1024
    res["hbm_bytes_per_launch"] = res["fetch_bytes_corrected_x2"] + res["write_bytes"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters:
    # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    res["mfma_busy_frac"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / (counters["GRBM_GUI_ACTIVE"] / 8 * 1024)
    res["clock_ghz"] = counters["GRBM_GUI_ACTIVE"] / 8 / (res["avg_duration_us_under_pmc"] * 1e3)
if "TCC_HIT_sum" in counters:
    res["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
