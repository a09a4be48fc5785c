#!/bin/bash
# conv1x1: minimum wave count below which the staged kernel takes the launch (ICM_1X1_MIN_WAVES)
set -e
mkdir -p gpurun_out
for mw in 1024 512 256; do
  echo "== ICM_1X1_MIN_WAVES=$mw" >> gpurun_out/r3_mw.txt
  timeout -k 10 300 env ICM_1X1_MIN_WAVES=$mw python tools/conv1x1_probe.py 2>/dev/null | grep "16 \|@16" >> gpurun_out/r3_mw.txt || exit 1
done
cat gpurun_out/r3_mw.txt
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-shape-table"
for mw in 1024 512; do
timeout -k 10 300 env ICM_1X1_MIN_WAVES=$mw $B > gpurun_out/r3_mw_$mw.log 2>/dev/null
python - <<PY
import json
l=[x for x in open("gpurun_out/r3_mw_$mw.log") if x.startswith("{")][-1]
j=json.loads(l); print("$mw", round(j["value"],1), round(j["ms_per_step"],3), "fwd", round(j["forward"]["value"],1), "stf", {k:(round(v["value"],1) if isinstance(v,dict) and "value" in v else None) for k,v in j["stf"].items()})
PY
done
