set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -k "wgrad or bench_size" > $O/t_pipe.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_pipe.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_pipe.log | head -20
for pp in 0 1; do echo "== ICM_WG_PIPE=$pp"; ICM_WG_PIPE=$pp run 300 python tools/tune_wgrad.py "x" 2>&1 | grep -E "5x5|gelu|3x3 480|1x1 192->192 @64|3x3 96" | cut -c1-160; done
for pp in 0 1 0 1; do ICM_WG_PIPE=$pp run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench10_$pp.json 2> $O/bench10.err; echo "pipe $pp: $(tail -1 $O/bench10_$pp.json | cut -c75-110)"; done
