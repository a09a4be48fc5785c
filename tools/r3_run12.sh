#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_bwd or winograd or split or gdn" > gpurun_out/r3_g1.log 2>&1 \
 && run ICM_PACK_GRID=128 > gpurun_out/r3_b_pg128.json 2>/dev/null \
 && run ICM_X=1 > gpurun_out/r3_b_pgauto.json 2>/dev/null \
 && run ICM_PACK_GRID=1024 > gpurun_out/r3_b_pg1024.json 2>/dev/null \
 && run ICM_PACK_GRID=256 > gpurun_out/r3_b_pg256.json 2>/dev/null
echo rc=$?; tail -2 gpurun_out/r3_g1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_pg*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
