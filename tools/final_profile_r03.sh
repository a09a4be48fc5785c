# end-of-round evidence (round 3): headline bench (+ forward / stf sub-objects, cpu baseline, full shape tables), rocprofv3
# --kernel-trace --stats of the same command (default = weight gradients on the side stream, and serialised), PMC passes
# of the family launches incl. the Winograd kernels -> gpurun_out/final3/   (copy the summaries into profiles/r03_*)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final3; rm -rf $O; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
ICM_SHAPE_TABLE=$O/shapes_train.jsonl run 500 python bench.py > $O/bench.json 2> $O/bench.err || true; tail -1 $O/bench.json | cut -c1-200
ICM_SHAPE_TABLE=$O/shapes_fwd.jsonl run 300 python bench.py --fwd-only --no-cpu-baseline > $O/bench_fwd.json 2>/dev/null || true; tail -1 $O/bench_fwd.json | cut -c1-160
ICM_SHAPE_TABLE=$O/shapes_stf.jsonl run 300 python bench.py --model stf --no-cpu-baseline > $O/bench_stf.json 2>/dev/null || true; tail -1 $O/bench_stf.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-shape-table --no-extras > $O/prof.log 2>&1 || true
tail -1 $O/prof.log | cut -c1-160
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
ICM_WG_EVERY=-1 run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-shape-table --no-extras > $O/prof_serial.log 2>&1 || true
tail -1 $O/prof_serial.log | cut -c1-160
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_serial.csv 2>/dev/null; rm -rf $O/prof
python3 $R/tools/kernel_stats_families.py $O/kernel_stats.csv 14 > $O/kernel_families_concurrent.json
python3 $R/tools/kernel_stats_families.py $O/kernel_stats_serial.csv 14 > $O/kernel_families_serial.json
cat $O/kernel_families_serial.json
cd $R && bash tools/pmc_family.sh > $O/pmc.log 2>&1; python3 tools/pmc_family_summary.py $O/pmc_family.json | cut -c1-900
rm -rf $R/gpurun_out/pmcf_*/*/*.db 2>/dev/null
