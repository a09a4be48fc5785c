#!/bin/bash
# round 3, GPU call 1: new-piece tests, cnn parity (small + bench geometry), A/B of the split slice section.
# Steps are chained with &&: after a failing GPU step nothing else touches the GPU in this call.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "split or grouped_conv_rejects or differentiable or conv_fwd_bwd or wgrad_every" > gpurun_out/r3_t1.log 2>&1 \
 && tail -3 gpurun_out/r3_t1.log \
 && timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q -s > gpurun_out/r3_t2.log 2>&1 \
 && tail -5 gpurun_out/r3_t2.log \
 && ICM_SLICE_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_split0.json 2> gpurun_out/r3_b_split0.err \
 && ICM_SLICE_SPLIT=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_split1.json 2> gpurun_out/r3_b_split1.err
rc=$?
echo "chain rc=$rc"
tail -15 gpurun_out/r3_t1.log; tail -25 gpurun_out/r3_t2.log 2>/dev/null
python - <<'PY'
import json
for t in ("0","1"):
    try:
        d=json.loads(open(f"gpurun_out/r3_b_split{t}.json").read().strip().splitlines()[-1])
        print("split",t,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms", d.get("last_step"))
    except Exception as e:
        print("split",t,"failed",e)
PY
exit $rc
