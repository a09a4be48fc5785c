set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -k "wgrad" > $O/t_t7.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_t7.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_t7.log | head -30
run 300 python tools/tune_wgrad.py "3x3 96" 2>&1 | grep -E "3x3" | cut -c1-170
run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench12.json 2> $O/bench12.err; tail -1 $O/bench12.json | cut -c1-150
