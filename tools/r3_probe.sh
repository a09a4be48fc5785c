#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for dbg in 7 15 23 31; do echo "== ICM_WINO_DEBUG=$dbg (1 no patch loads, 2 no weight loads, 4 no epilogue, 8 no transform/store, 16 no B reads)"; ICM_WINO_DEBUG=$dbg PROBE_ALGOS=1 timeout -k 10 200 python tools/wino_probe.py || exit 1; done
} > gpurun_out/r3_probe2.txt 2>&1
echo rc=$?
