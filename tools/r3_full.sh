#!/bin/bash
# full GPU suite + smoke
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1 || { tail -20 gpurun_out/r3_smoke.log; exit 1; }
tail -2 gpurun_out/r3_smoke.log
