#!/bin/bash
# thin-output end_conv of stf: parity tests + same-box A/B (ICM_THIN_OUT)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stf.py tests/test_gpu_b16_stf.py -x -q -k "thin or stf" > gpurun_out/r3_thin_tests.log 2>&1 || { tail -40 gpurun_out/r3_thin_tests.log; exit 1; }
tail -2 gpurun_out/r3_thin_tests.log
B="python bench.py --model stf --steps 20 --warmup 5 --no-cpu-baseline --no-shape-table"
timeout -k 10 300 env ICM_THIN_OUT=0 $B > gpurun_out/r3_thin_a.log 2>/dev/null && \
timeout -k 10 300 $B > gpurun_out/r3_thin_b.log 2>/dev/null && \
timeout -k 10 300 env ICM_THIN_OUT=0 $B --fwd-only > gpurun_out/r3_thin_c.log 2>/dev/null && \
timeout -k 10 300 $B --fwd-only > gpurun_out/r3_thin_d.log 2>/dev/null
for f in a b c d; do python - <<PY
import json
l=[x for x in open("gpurun_out/r3_thin_$f.log") if x.startswith("{")][-1]
j=json.loads(l); print("$f", j["metric"][-30:], round(j["value"],1), round(j["ms_per_step"],3))
PY
done
