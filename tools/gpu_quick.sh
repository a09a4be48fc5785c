set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -k "${1:-wgrad or conv}" > $O/t_quick.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_quick.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_quick.log | head -30
ICM_SHAPE_TABLE=$O/shapes_q.jsonl run 300 python bench.py --no-cpu-baseline > $O/bench_q.json 2> $O/bench_q.err; tail -1 $O/bench_q.json | cut -c1-150
run 300 python bench.py --no-cpu-baseline --fwd-only --no-shape-table > $O/bench_qf.json 2>> $O/bench_q.err; tail -1 $O/bench_qf.json | cut -c1-150
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench_q.json").read().strip().splitlines()[-1])
for k,v in d["roofline_families"].items(): print(k, v)
PY
