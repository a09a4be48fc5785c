#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
run ICM_X=0 > gpurun_out/r3_b_c_base.json 2>/dev/null \
 && run ICM_HOLD_CHAIN_WGRADS=1 > gpurun_out/r3_b_c_hold.json 2>/dev/null \
 && run ICM_WG_EVERY=16 > gpurun_out/r3_b_c_every16.json 2>/dev/null \
 && run ICM_WG_EVERY=32 ICM_WG_MIN=16 > gpurun_out/r3_b_c_every32.json 2>/dev/null \
 && run ICM_WG_EVERY=-1 > gpurun_out/r3_b_c_serial.json 2>/dev/null \
 && run ICM_PACK_WINDOW=48 > gpurun_out/r3_b_c_pw48.json 2>/dev/null \
 && run ICM_PACK_WINDOW=12 > gpurun_out/r3_b_c_pw12.json 2>/dev/null \
 && run ICM_MAT_MIN_PIXELS=512 > gpurun_out/r3_b_c_mat512.json 2>/dev/null
echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_c_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
