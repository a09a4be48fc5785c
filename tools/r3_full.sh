#!/bin/bash
# full GPU suite + probe + bench line with extras (chained)
set -o pipefail
mkdir -p gpurun_out
true \
 && timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_full_tests.log 2>&1 \
 && timeout -k 10 400 python bench.py --steps 20 > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err
rc=$?
echo "chain rc=$rc"
grep -v amdgpu.ids gpurun_out/r3_probe3.txt
tail -15 gpurun_out/r3_full_tests.log
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_full.json").read().strip().splitlines()[-1])
    print("train",round(d["value"],1),"img/s", "fwd", d.get("forward",{}).get("value"), "stf", d.get("stf"))
    print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("bench failed", e)
PY
exit $rc
