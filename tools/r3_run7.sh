#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd_wgrad or wgrad_every" > gpurun_out/r3_q1.log 2>&1 \
 && { echo "== ICM_WW_PRE=1"; ICM_WW_PRE=1 timeout -k 10 300 python tools/wgrad_probe.py; echo "== ICM_WW_PRE=0"; ICM_WW_PRE=0 timeout -k 10 300 python tools/wgrad_probe.py; } > gpurun_out/r3_wgprobe2.txt 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -k "oracle_small or b16_trainer" > gpurun_out/r3_q2.log 2>&1 \
 && run ICM_WW_PRE=0 > gpurun_out/r3_b_wwpre0.json 2> gpurun_out/r3_b_wwpre0.err \
 && run ICM_WW_PRE=1 > gpurun_out/r3_b_wwpre1.json 2> gpurun_out/r3_b_wwpre1.err \
 && run ICM_WW_PRE=1 ICM_WINO_WG_BATCH=8 > gpurun_out/r3_b_wwpre1_b8.json 2> gpurun_out/r3_b_wwpre1_b8.err
rc=$?
echo "chain rc=$rc"; tail -4 gpurun_out/r3_q1.log; grep -v amdgpu.ids gpurun_out/r3_wgprobe2.txt; tail -4 gpurun_out/r3_q2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_wwpre*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
exit $rc
