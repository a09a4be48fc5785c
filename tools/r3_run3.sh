#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd or split" > gpurun_out/r3_x1.log 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py -x -q > gpurun_out/r3_x2.log 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_b16.py -x -q -k "eval or trainer" > gpurun_out/r3_x3.log 2>&1 \
 && for v in 0 1.0e8 2.0e8 4.0e8 1.0e9; do ICM_WINO_MIN_WORK=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_mw_$v.json 2> gpurun_out/r3_b_mw_$v.err || exit 1; done \
 && ICM_SHAPE_TABLE=gpurun_out/r3_shapes_wino2.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > gpurun_out/r3_b_wino2.json 2> gpurun_out/r3_b_wino2.err \
 && ICM_SLICE_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_wino2_split0.json 2> gpurun_out/r3_b_wino2_split0.err
rc=$?
echo "chain rc=$rc"
tail -5 gpurun_out/r3_x1.log; tail -5 gpurun_out/r3_x2.log 2>/dev/null; tail -5 gpurun_out/r3_x3.log 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_mw_*.json"))+["gpurun_out/r3_b_wino2.json","gpurun_out/r3_b_wino2_split0.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e:
        print(f,"failed",e)
PY
exit $rc
