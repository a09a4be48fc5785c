set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -rA -k "codec or wgrad" > $O/t_wg3.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_wg3.log | tail -2
grep -E "^(FAILED|ERROR)|pmf max|y stream|actual .* bpp|inference:" $O/t_wg3.log | head -30
run 600 python tools/tune_wgrad.py 1x1 > $O/tune_wgrad3.txt 2>&1; cat $O/tune_wgrad3.txt | cut -c1-200
run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench5.json 2> $O/bench5.err; tail -1 $O/bench5.json | cut -c1-150
