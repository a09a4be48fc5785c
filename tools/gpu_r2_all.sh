# round-2 GPU pass: the whole -m gpu suite (no -x: every failure is wanted), then the headline bench (per-shape table
# included) and the forward-only line.  Outputs under gpurun_out/r2/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -rA --timeout 900 > $O/t_all.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_all.log | tail -2
grep -E "^(FAILED|ERROR)" $O/t_all.log | head -30
run 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-400
run 300 python bench.py --fwd-only --no-cpu-baseline > $O/bench_fwd.json 2>> $O/bench.err; tail -1 $O/bench_fwd.json | cut -c1-300
