# end-of-round evidence: GPU tests, headline bench (+ cpu baseline), rocprofv3 --kernel-trace --stats of the same
# command, PMC passes of the dominant kernel, forward-only and stf benches.  Outputs under gpurun_out/final/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
run() { timeout -k 10 900 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run python -m pytest tests -m gpu -q -p no:cacheprovider > $O/t_all.log 2>&1; tail -3 $O/t_all.log
run python bench.py > $O/bench.json 2> $O/bench.err || true; tail -1 $O/bench.json | cut -c1-300
run python bench.py --fwd-only --no-cpu-baseline > $O/bench_fwd.json 2>/dev/null || true; tail -1 $O/bench_fwd.json | cut -c1-200
run python bench.py --model stf --no-cpu-baseline > $O/bench_stf.json 2>/dev/null || true; tail -1 $O/bench_stf.json | cut -c1-200
run python bench.py --model stf --fwd-only --no-cpu-baseline > $O/bench_stf_fwd.json 2>/dev/null || true; tail -1 $O/bench_stf_fwd.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
run rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1 || true
tail -1 $O/prof.log | cut -c1-200
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
cd $R && bash tools/pmc_dominant.sh > $O/pmc.log 2>&1; python3 tools/pmc_dominant_summary.py $O/pmc_dominant.json | cut -c1-300
