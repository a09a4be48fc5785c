set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x > $O/t_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_full.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_full.log | head -20
ICM_SHAPE_TABLE=$O/shapes_train2.jsonl run 300 python bench.py --no-cpu-baseline > $O/bench7.json 2> $O/bench7.err; tail -1 $O/bench7.json | cut -c1-150
ICM_SHAPE_TABLE=$O/shapes_fwd2.jsonl run 300 python bench.py --no-cpu-baseline --fwd-only > $O/bench7_fwd.json 2>> $O/bench7.err; tail -1 $O/bench7_fwd.json | cut -c1-150
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench7.json").read().strip().splitlines()[-1])
for k,v in d["roofline_families"].items(): print(k, v)
PY
