#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or split" > gpurun_out/r3_e1.log 2>&1 \
 && { echo "== ICM_WINO8=1"; ICM_WINO8=1 timeout -k 10 200 python tools/wino_probe.py; echo "== ICM_WINO8=0"; ICM_WINO8=0 PROBE_ALGOS=1 timeout -k 10 200 python tools/wino_probe.py; for t in 2 3 4; do echo "== ICM_WINO8=1 ICM_WINO_TCO=$t"; ICM_WINO_TCO=$t PROBE_ALGOS=1 timeout -k 10 200 python tools/wino_probe.py; done; } > gpurun_out/r3_probe5.txt 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -k "oracle_small or b16_trainer or b16_eval" > gpurun_out/r3_e2.log 2>&1 \
 && run ICM_WINO8=0 > gpurun_out/r3_b_w8_0.json 2> gpurun_out/r3_b_w8_0.err \
 && run ICM_WINO8=1 > gpurun_out/r3_b_w8_1.json 2> gpurun_out/r3_b_w8_1.err \
 && run ICM_WINO8=1 ICM_WINO_MIN_WORK=1.0e8 > gpurun_out/r3_b_w8_1_mw1e8.json 2> gpurun_out/r3_b_w8_1_mw1e8.err \
 && run ICM_WINO8=1 ICM_WINO_MIN_CIN=64 > gpurun_out/r3_b_w8_1_cin64.json 2> gpurun_out/r3_b_w8_1_cin64.err \
 && run ICM_WINO8=1 ICM_SLICE_SPLIT=1 > gpurun_out/r3_b_w8_1_split.json 2> gpurun_out/r3_b_w8_1_split.err
rc=$?
echo "chain rc=$rc"; tail -4 gpurun_out/r3_e1.log; grep -v amdgpu.ids gpurun_out/r3_probe5.txt; tail -4 gpurun_out/r3_e2.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_w8_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
exit $rc
