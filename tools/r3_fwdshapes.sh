#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ICM_SHAPE_TABLE=gpurun_out/r3_shapes_fwd2.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --fwd-only --steps 20 > gpurun_out/r3_b_fwd2.json 2> gpurun_out/r3_b_fwd2.err \
 && ICM_SHAPE_TABLE=gpurun_out/r3_shapes_train2.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > gpurun_out/r3_b_train2.json 2> gpurun_out/r3_b_train2.err
echo rc=$?
