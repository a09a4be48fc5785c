set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 900 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return 0; }
run python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/t_ops.log 2>&1
tail -5 gpurun_out/t_ops.log
run python -m pytest tests/test_gpu_wacnn.py -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/t_wacnn.log 2>&1
tail -5 gpurun_out/t_wacnn.log
