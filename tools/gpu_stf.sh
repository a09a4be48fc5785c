set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
ICM_SHAPE_TABLE=$O/shapes_stf.jsonl run 300 python bench.py --no-cpu-baseline --model stf > $O/bench_stf.json 2> $O/bench_stf.err; tail -1 $O/bench_stf.json | cut -c1-150
ICM_SHAPE_TABLE=$O/shapes_stf_fwd.jsonl run 300 python bench.py --no-cpu-baseline --model stf --fwd-only > $O/bench_stf_fwd.json 2>> $O/bench_stf.err; tail -1 $O/bench_stf_fwd.json | cut -c1-150
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench_stf.json").read().strip().splitlines()[-1])
for k,v in d["roofline_families"].items(): print(k, v)
for r in d["roofline_shapes"][:14]: print(r["ms"], r["launches"], r["tflops"], r["shape"])
PY
