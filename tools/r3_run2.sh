#!/bin/bash
# round 3: Winograd kernel -- unit tests, model parity, A/B bench (chained: nothing runs after a failing GPU step)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd" > gpurun_out/r3_w1.log 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r3_w2.log 2>&1 \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -s > gpurun_out/r3_w3.log 2>&1 \
 && ICM_WINO=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_wino0.json 2> gpurun_out/r3_b_wino0.err \
 && ICM_WINO=1 ICM_SHAPE_TABLE=gpurun_out/r3_shapes_wino1.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r3_b_wino1.json 2> gpurun_out/r3_b_wino1.err \
 && ICM_WINO=1 ICM_SLICE_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_wino1_split0.json 2> gpurun_out/r3_b_wino1_split0.err
rc=$?
echo "chain rc=$rc"
tail -25 gpurun_out/r3_w1.log; tail -8 gpurun_out/r3_w2.log 2>/dev/null; tail -12 gpurun_out/r3_w3.log 2>/dev/null
python - <<'PY'
import json
for t in ("wino0","wino1","wino1_split0"):
    try:
        d=json.loads(open(f"gpurun_out/r3_b_{t}.json").read().strip().splitlines()[-1])
        print(t,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms", d.get("last_step"))
    except Exception as e:
        print(t,"failed",e)
PY
exit $rc
