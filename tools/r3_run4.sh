#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
true \
 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -k "oracle_small or b16_trainer" > gpurun_out/r3_y2.log 2>&1 \
 && ICM_WINO_WGRAD=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_ww0.json 2> gpurun_out/r3_b_ww0.err \
 && ICM_WINO_WGRAD=1 ICM_SHAPE_TABLE=gpurun_out/r3_shapes_ww1.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r3_b_ww1.json 2> gpurun_out/r3_b_ww1.err
rc=$?
echo "chain rc=$rc"
tail -8 gpurun_out/r3_y1.log; grep -v amdgpu.ids gpurun_out/r3_wgprobe.txt; tail -5 gpurun_out/r3_y2.log 2>/dev/null
python - <<'PY'
import json
for t in ("ww0","ww1"):
    try:
        d=json.loads(open(f"gpurun_out/r3_b_{t}.json").read().strip().splitlines()[-1])
        print(t,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e:
        print(t,"failed",e)
PY
exit $rc
