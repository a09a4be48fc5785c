"""Summarise tools/pmc_family.sh (gpurun_out/pmcf_*) into profiles/r03_pmc_family.json.
HBM-side bytes per launch = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KB
in the counter output; separate passes per counter set; averages over the launches of the named kernel."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_pmc_family.json")
WHAT = {
    "wg5": ("wgrad", "wgrad_", "wgrad 5x5 s2 192->192 on [16,192,128,128] (g_a.2 / g_s.6)",
            4 * (16 * 192 * 128 * 128 + 16 * 192 * 64 * 64 + 192 * 192 * 25), 2.0 * 16 * 192 * 192 * 25 * 64 * 64),
    "wg1": ("wgrad_1x1", "wgrad_", "wgrad 1x1 192->192 on [16,192,64,64]",
            4 * (2 * 16 * 192 * 64 * 64 + 192 * 192), 2.0 * 16 * 192 * 192 * 64 * 64),
    "conv": ("conv", "conv_igemm_kernel", "g_a.2 forward conv5x5 s2 192->192 on [16,192,128,128]",
             4 * (16 * 192 * 128 * 128 + 16 * 192 * 64 * 64 + 192 * 192 * 25), 2.0 * 16 * 192 * 192 * 25 * 64 * 64),
    # Winograd kernels (round 3): algorithmic bytes / FLOP of the DIRECT form of the same problem (what the launch
    # computes); the kernel executes 4/9 of the multiply-adds and moves the 16-point weights / transformed operands
    "wino": ("conv_wino", "conv_wino", "slice-chain second layer conv3x3 224->176 on [16,224,16,16] x 10 members (Winograd F(2x2,3x3))",
             10 * 4 * (16 * 224 * 256 + 16 * 176 * 256 + 176 * 224 * 9), 2.0 * 9 * 10 * 16 * 256 * 224 * 176),
    "wwino": ("wgrad_wino", "wgrad_wino_kernel", "slice-chain first-layer wgrad 3x3 512->224 on [16,512,16,16] x 5 problems (Winograd)",
              5 * 4 * (16 * 512 * 256 + 16 * 224 * 256 + 224 * 512 * 9), 2.0 * 9 * 5 * 16 * 256 * 512 * 224),
}
res = {}
for tag, (key, kmatch, workload, alg, flop) in WHAT.items():
    counters, durs, kname = {}, [], None
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcf_{tag}_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            acc, cnt = {}, {}
            for r in csv.DictReader(open(f)):
                if kmatch not in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
                    continue
                kname = r["Kernel_Name"]
                c = r["Counter_Name"]
                acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
                cnt[c] = cnt.get(c, 0) + 1
            for c in acc:
                counters[c] = acc[c] / cnt[c]
        for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
            for r in csv.DictReader(open(f)):
                if kmatch in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if not counters:
        continue
    e = {"kernel": kname, "workload": workload, "avg_duration_us_under_pmc": sum(durs) / max(len(durs), 1),
         "counters": counters, "algorithmic_bytes_per_launch": alg, "algorithmic_flop_per_launch": flop}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        e["fetch_bytes_corrected_x2"] = counters["FETCH_SIZE"] * 1024 * 2
        e["write_bytes"] = counters["WRITE_SIZE"] * 1024
        e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected_x2"] + e["write_bytes"]
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / alg
    if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters:
        e["mfma_busy_frac"] = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / (counters["GRBM_GUI_ACTIVE"] / 8 * 1024)
        e["clock_ghz"] = counters["GRBM_GUI_ACTIVE"] / 8 / (e["avg_duration_us_under_pmc"] * 1e3)
    if "TCC_HIT_sum" in counters:
        e["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
    res[key + "_detail"] = e
    if "hbm_bytes_per_launch" in e:
        res[key] = e["hbm_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:1500])
