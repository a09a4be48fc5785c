#!/bin/bash
# per-shape tables of the training step with and without the split slice section (same box)
set -o pipefail
mkdir -p gpurun_out
ICM_SLICE_SPLIT=0 ICM_SHAPE_TABLE=gpurun_out/r3_shapes_split0.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > gpurun_out/r3_bs0.json 2> gpurun_out/r3_bs0.err \
 && ICM_SLICE_SPLIT=1 ICM_SHAPE_TABLE=gpurun_out/r3_shapes_split1.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > gpurun_out/r3_bs1.json 2> gpurun_out/r3_bs1.err
echo rc=$?
