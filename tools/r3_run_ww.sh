#!/bin/bash
# same-box A/B after the wgrad_wino rewrite: dword vs vector gathers, Winograd weight gradients also for the 64x64 ResidualUnit layers
set -e
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-shape-table --no-extras"
timeout -k 10 300 env ICM_WW_NOVEC=1 $B > gpurun_out/r3_ww_a.log 2> gpurun_out/r3_ww_a.err && \
timeout -k 10 300 $B > gpurun_out/r3_ww_b.log 2> gpurun_out/r3_ww_b.err && \
timeout -k 10 300 env ICM_WINO_WG_MAX_PIXELS=100000 ICM_WINO_WG_MIN_C=96 $B > gpurun_out/r3_ww_c.log 2> gpurun_out/r3_ww_c.err && \
timeout -k 10 300 env ICM_WINO_WG_MAX_PIXELS=100000 ICM_WINO_WG_MIN_C=64 $B > gpurun_out/r3_ww_d.log 2> gpurun_out/r3_ww_d.err && \
timeout -k 10 300 $B > gpurun_out/r3_ww_e.log 2> gpurun_out/r3_ww_e.err
for f in a b c d e; do python - <<PY
import json
l=[x for x in open("gpurun_out/r3_ww_$f.log") if x.startswith("{")][-1]
j=json.loads(l); print("$f", j["value"], j["ms_per_step"])
PY
done
