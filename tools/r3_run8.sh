#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_graph.py -x -q > gpurun_out/r3_d1.log 2>&1 \
 && timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20 > gpurun_out/r3_b_nojoin.json 2> gpurun_out/r3_b_nojoin.err
rc=$?
echo "chain rc=$rc"; tail -4 gpurun_out/r3_d1.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_b_nojoin.json").read().strip().splitlines()[-1]); print(round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
PY
[ $rc -eq 0 ] && bash tools/final_profile_r03.sh
