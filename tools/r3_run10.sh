#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 200 python tools/wino_probe.py > gpurun_out/r3_probe6.txt 2>&1 \
 && timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd" > gpurun_out/r3_f1.log 2>&1 \
 && run ICM_SLICE_SPLIT=0 > gpurun_out/r3_b_fin_u.json 2> gpurun_out/r3_b_fin_u.err \
 && run ICM_SLICE_SPLIT=1 > gpurun_out/r3_b_fin_s.json 2> gpurun_out/r3_b_fin_s.err \
 && run ICM_SLICE_SPLIT=1 ICM_KSPLIT_A=0 ICM_KSPLIT_B=0 > gpurun_out/r3_b_fin_s_ks0.json 2> gpurun_out/r3_b_fin_s_ks0.err \
 && ICM_SLICE_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-shape-table --fwd-only --steps 30 > gpurun_out/r3_b_fin_u_fwd.json 2>/dev/null \
 && ICM_SLICE_SPLIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-shape-table --fwd-only --steps 30 > gpurun_out/r3_b_fin_s_fwd.json 2>/dev/null \
 && run ICM_SLICE_SPLIT=0 ICM_WINO8_MINWG=96 > gpurun_out/r3_b_fin_u_min96.json 2> /dev/null \
 && run ICM_SLICE_SPLIT=0 ICM_WINO_WG_BATCH=2 > gpurun_out/r3_b_fin_u_wgb2.json 2> /dev/null
rc=$?
echo "chain rc=$rc"; grep -v amdgpu.ids gpurun_out/r3_probe6.txt; tail -3 gpurun_out/r3_f1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_fin_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
exit $rc
