#!/bin/bash
# same-box A/B: vector gathers in the Winograd input transform launches (ICM_WINO_XF_NOVEC=1 = dword gathers)
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "wino" > gpurun_out/r3_xf_tests.log 2>&1 || { tail -30 gpurun_out/r3_xf_tests.log; exit 1; }
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-shape-table"
timeout -k 10 300 env ICM_WINO_XF_NOVEC=1 $B > gpurun_out/r3_xf_a.log 2> gpurun_out/r3_xf_a.err && \
timeout -k 10 300 $B > gpurun_out/r3_xf_b.log 2> gpurun_out/r3_xf_b.err && \
timeout -k 10 300 env ICM_WINO_XF_NOVEC=1 $B > gpurun_out/r3_xf_c.log 2> gpurun_out/r3_xf_c.err && \
timeout -k 10 300 $B > gpurun_out/r3_xf_d.log 2> gpurun_out/r3_xf_d.err
tail -2 gpurun_out/r3_xf_tests.log
for f in a b c d; do python - <<PY
import json
l=[x for x in open("gpurun_out/r3_xf_$f.log") if x.startswith("{")][-1]
j=json.loads(l); print("$f", round(j["value"],1), round(j["ms_per_step"],3), "fwd", round(j["forward"]["value"],1), "stf", round(j["stf"]["value"],1))
PY
done
