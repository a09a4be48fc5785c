#!/bin/bash
# same-box A/B: Winograd channel threshold (ICM_WINO_MIN_CIN) after the input-transform speed-up
set -e
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-shape-table"
timeout -k 10 300 $B > gpurun_out/r3_mc_a.log 2> gpurun_out/r3_mc_a.err && \
timeout -k 10 300 env ICM_WINO_MIN_CIN=96 $B > gpurun_out/r3_mc_b.log 2> gpurun_out/r3_mc_b.err && \
timeout -k 10 300 env ICM_WINO_MIN_CIN=64 $B > gpurun_out/r3_mc_c.log 2> gpurun_out/r3_mc_c.err && \
timeout -k 10 300 $B > gpurun_out/r3_mc_d.log 2> gpurun_out/r3_mc_d.err && \
timeout -k 10 300 env ICM_WINO_MIN_CIN=96 $B > gpurun_out/r3_mc_e.log 2> gpurun_out/r3_mc_e.err
for f in a b c d e; do python - <<PY
import json
l=[x for x in open("gpurun_out/r3_mc_$f.log") if x.startswith("{")][-1]
j=json.loads(l); print("$f", round(j["value"],1), round(j["ms_per_step"],3), "fwd", round(j["forward"]["value"],1), "stf", {k:(round(v["value"],1) if isinstance(v,dict) and "value" in v else None) for k,v in j["stf"].items()})
PY
done
