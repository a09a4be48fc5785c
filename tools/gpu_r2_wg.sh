set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -k "codec or wgrad or bench_size or rd_loss" > $O/t_wg.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_wg.log | tail -2
grep -E "^(FAILED|ERROR)" $O/t_wg.log | head -30
run 600 python tools/tune_wgrad.py > $O/tune_wgrad.txt 2>&1; cat $O/tune_wgrad.txt | cut -c1-200
run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench3.json 2> $O/bench3.err; tail -1 $O/bench3.json | cut -c1-150
