"""Per-step family totals from a rocprofv3 kernel_stats.csv: python tools/kernel_stats_families.py <csv> <steps+warmup+1>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = {}
for r in rows:
    n, t, c = r["Name"], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    if "wgrad" in n:
        k = "wgrad_* kernels + wgrad_reduce_kernel"
    elif "conv_igemm" in n or "conv1x1" in n or "conv_ks8" in n or "conv_wino" in n or "wino_input_transform" in n:
        k = "conv_igemm_kernel + conv1x1_kernel + conv_ks8_kernel + conv_wino_kernel (+ wino_input_transform)"
    elif "winattn" in n:
        k = "winattn kernels"
    elif "pack_weights" in n:
        k = "pack_weights kernels"
    else:
        k = "other"
    e = fam.setdefault(k, {"ms_per_step": 0.0, "launches_per_step": 0.0})
    e["ms_per_step"] += t / nsteps
    e["launches_per_step"] += c / nsteps
for e in fam.values():
    e["ms_per_step"] = round(e["ms_per_step"], 3)
    e["launches_per_step"] = round(e["launches_per_step"], 1)
    e["avg_launch_ms"] = round(e["ms_per_step"] / max(e["launches_per_step"], 1e-9), 4)
print(json.dumps({"steps_in_trace": nsteps, "families": fam, "kernel_ms_per_step": round(sum(e["ms_per_step"] for e in fam.values()), 3)}))
