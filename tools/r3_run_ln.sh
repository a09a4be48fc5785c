#!/bin/bash
# LayerNorm: fused parameter gradients + cached C=384 kernels: parity tests + same-box A/B on the stf step (ICM_LN_NOFUSE)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_stf.py tests/test_gpu_b16_stf.py -x -q > gpurun_out/r3_ln_tests.log 2>&1 || { tail -40 gpurun_out/r3_ln_tests.log; exit 1; }
tail -2 gpurun_out/r3_ln_tests.log
B="python bench.py --model stf --steps 20 --warmup 5 --no-cpu-baseline --no-shape-table"
timeout -k 10 300 env ICM_LN_NOFUSE=1 $B > gpurun_out/r3_ln_a.log 2>/dev/null && \
timeout -k 10 300 $B > gpurun_out/r3_ln_b.log 2>/dev/null && \
timeout -k 10 300 env ICM_LN_NOFUSE=1 $B > gpurun_out/r3_ln_c.log 2>/dev/null && \
timeout -k 10 300 $B > gpurun_out/r3_ln_d.log 2>/dev/null && \
timeout -k 10 300 $B --fwd-only > gpurun_out/r3_ln_e.log 2>/dev/null
for f in a b c d e; do python - <<PY
import json
l=[x for x in open("gpurun_out/r3_ln_$f.log") if x.startswith("{")][-1]
j=json.loads(l); print("$f", j["metric"][-30:], round(j["value"],1), round(j["ms_per_step"],3))
PY
done
