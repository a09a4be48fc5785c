#!/bin/bash
# conv1x1 ablation (ICM_1X1_DEBUG: 1 no B loads, 2 no A loads, 4 no epilogue)
set -e
mkdir -p gpurun_out
for cfg in 0 1 2 4 5 7; do
  echo "== ICM_1X1_DEBUG=$cfg" >> gpurun_out/r3_1x1.txt
  timeout -k 10 300 env ICM_1X1_DEBUG=$cfg python tools/conv1x1_probe.py 2>/dev/null >> gpurun_out/r3_1x1.txt || exit 1
done
cat gpurun_out/r3_1x1.txt
