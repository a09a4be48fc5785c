#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --no-shape-table --steps 20; }
timeout -k 10 900 python -m pytest tests/test_gpu_wacnn.py tests/test_gpu_b16.py tests/test_gpu_stf.py -x -q -k "oracle_small or b16_trainer or stf_train_grads" > gpurun_out/r3_z1.log 2>&1 \
 && run ICM_KSPLIT_A=0 ICM_KSPLIT_B=0 > gpurun_out/r3_b_ks00.json 2> gpurun_out/r3_b_ks00.err \
 && run ICM_KSPLIT_A=4 ICM_KSPLIT_B=0 > gpurun_out/r3_b_ks40.json 2> gpurun_out/r3_b_ks40.err \
 && run ICM_KSPLIT_A=4 ICM_KSPLIT_B=1 > gpurun_out/r3_b_ks41.json 2> gpurun_out/r3_b_ks41.err \
 && run ICM_KSPLIT_A=8 ICM_KSPLIT_B=1 > gpurun_out/r3_b_ks81.json 2> gpurun_out/r3_b_ks81.err \
 && run ICM_KSPLIT_A=4 ICM_KSPLIT_B=1 ICM_SLICE_SPLIT=0 > gpurun_out/r3_b_ks41_unsplit.json 2> gpurun_out/r3_b_ks41_unsplit.err \
 && run ICM_KSPLIT_A=4 ICM_KSPLIT_B=1 ICM_WINO_MIN_WORK=1.0e8 > gpurun_out/r3_b_ks41_mw1e8.json 2> gpurun_out/r3_b_ks41_mw1e8.err
rc=$?
echo "chain rc=$rc"; tail -4 gpurun_out/r3_z1.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_ks*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f,round(d["value"],1),"img/s",round(d["ms_per_step"],2),"ms")
    except Exception as e: print(f,"failed",e)
PY
exit $rc
