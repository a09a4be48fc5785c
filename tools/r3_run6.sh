#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or split" > gpurun_out/r3_p1.log 2>&1 \
 && { echo "== ICM_WINO_PRE=1"; ICM_WINO_PRE=1 timeout -k 10 200 python tools/wino_probe.py; echo "== ICM_WINO_PRE=0"; ICM_WINO_PRE=0 PROBE_ALGOS=1 timeout -k 10 200 python tools/wino_probe.py; } > gpurun_out/r3_probe4.txt 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -k "oracle_small or b16_trainer or b16_eval" > gpurun_out/r3_p2.log 2>&1 \
 && run ICM_WINO_PRE=0 ICM_SLICE_SPLIT=0 > gpurun_out/r3_b_pre0_u.json 2> gpurun_out/r3_b_pre0_u.err \
 && run ICM_WINO_PRE=1 ICM_SLICE_SPLIT=0 > gpurun_out/r3_b_pre1_u.json 2> gpurun_out/r3_b_pre1_u.err \
 && run ICM_WINO_PRE=1 ICM_SLICE_SPLIT=1 > gpurun_out/r3_b_pre1_s.json 2> gpurun_out/r3_b_pre1_s.err \
 && run ICM_WINO_PRE=1 ICM_SLICE_SPLIT=0 ICM_WINO_MIN_WORK=1.0e8 > gpurun_out/r3_b_pre1_u_mw1e8.json 2> gpurun_out/r3_b_pre1_u_mw1e8.err \
 && run ICM_WINO_PRE=1 ICM_SLICE_SPLIT=0 ICM_WINO_MIN_WORK=5.0e7 > gpurun_out/r3_b_pre1_u_mw5e7.json 2> gpurun_out/r3_b_pre1_u_mw5e7.err
rc=$?
echo "chain rc=$rc"; tail -4 gpurun_out/r3_p1.log; grep -v amdgpu.ids gpurun_out/r3_probe4.txt; tail -4 gpurun_out/r3_p2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_pre*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
exit $rc
