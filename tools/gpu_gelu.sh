set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/t_gelu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_gelu.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_gelu.log | head -20
run 300 python tools/tune_wgrad.py "gelu" 2>&1 | grep -E "gelu" | cut -c1-160
ICM_SHAPE_TABLE=$O/shapes_train3.jsonl run 300 python bench.py --no-cpu-baseline > $O/bench11.json 2> $O/bench11.err; tail -1 $O/bench11.json | cut -c1-150
ICM_SHAPE_TABLE=$O/shapes_fwd3.jsonl run 300 python bench.py --no-cpu-baseline --fwd-only > $O/bench11_fwd.json 2>> $O/bench11.err; tail -1 $O/bench11_fwd.json | cut -c1-150
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench11.json").read().strip().splitlines()[-1])
for k,v in d["roofline_families"].items(): print(k, v)
PY
