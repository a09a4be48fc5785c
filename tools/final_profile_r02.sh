# end-of-round evidence (round 2): GPU tests, headline bench (+ cpu baseline, full shape tables), rocprofv3 --kernel-trace
# --stats of the same command, PMC passes of the family launches, forward-only and stf benches -> gpurun_out/final2/
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2; rm -rf $O; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
if [ "$1" != "notests" ]; then run 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/t_all.log 2>&1; tail -3 $O/t_all.log; fi
ICM_SHAPE_TABLE=$O/shapes_train.jsonl run 400 python bench.py > $O/bench.json 2> $O/bench.err || true; tail -1 $O/bench.json | cut -c1-200
ICM_SHAPE_TABLE=$O/shapes_fwd.jsonl run 300 python bench.py --fwd-only --no-cpu-baseline > $O/bench_fwd.json 2>/dev/null || true; tail -1 $O/bench_fwd.json | cut -c1-160
ICM_SHAPE_TABLE=$O/shapes_stf.jsonl run 300 python bench.py --model stf --no-cpu-baseline > $O/bench_stf.json 2>/dev/null || true; tail -1 $O/bench_stf.json | cut -c1-160
run 300 python bench.py --model stf --fwd-only --no-cpu-baseline --no-shape-table > $O/bench_stf_fwd.json 2>/dev/null || true; tail -1 $O/bench_stf_fwd.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-shape-table > $O/prof.log 2>&1 || true
tail -1 $O/prof.log | cut -c1-160
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
cd $R && bash tools/pmc_family.sh > $O/pmc.log 2>&1; python3 tools/pmc_family_summary.py $O/pmc_family.json | cut -c1-600
rm -rf $R/gpurun_out/pmcf_*/*/*.db 2>/dev/null
