set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 1000 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run python -m pytest tests/test_gpu_ops.py::test_rd_loss_and_adam tests/test_gpu_wacnn.py::test_trainer_two_steps_vs_oracle -m gpu -q -s --timeout 600 -p no:cacheprovider > gpurun_out/t_trainer.log 2>&1
tail -6 gpurun_out/t_trainer.log
run python bench.py --steps 5 --warmup 2 > gpurun_out/bench1.log 2>&1 || true
tail -3 gpurun_out/bench1.log
cd /tmp && export TMPDIR=/tmp
run rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1 || true
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
ls $GRAFT_REPO_ROOT/gpurun_out/prof1 | head
