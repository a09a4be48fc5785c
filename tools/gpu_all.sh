set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 1000 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/t_all.log 2>&1
tail -4 gpurun_out/t_all.log
run python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1 || true
tail -1 gpurun_out/bench.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof
run rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 || true
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof.log | cut -c1-300
