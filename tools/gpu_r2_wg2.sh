set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_codec.py tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -rA -k "tables or wgrad" > $O/t_wg2.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_wg2.log | tail -2
grep -E "^(FAILED|ERROR)|GC cdf" $O/t_wg2.log | head -30
run 600 python tools/tune_wgrad.py 1x1 > $O/tune_wgrad2.txt 2>&1; cat $O/tune_wgrad2.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_wg
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wg -- python3 $R/tools/tune_wgrad.py "192->192 @64" > $O/prof_wg.log 2>&1
f=$(find $O/prof_wg -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-200; cp "$f" $O/prof_wg_stats.csv; rm -rf $O/prof_wg
cd $R
run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench4.json 2> $O/bench4.err; tail -1 $O/bench4.json | cut -c1-150
