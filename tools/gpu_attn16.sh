set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_stf.py -q -p no:cacheprovider --timeout 600 -k "attention or gate or swin or stf" > $O/t_attn16.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_attn16.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_attn16.log | head -30
run 300 python bench.py --no-cpu-baseline --model stf > $O/bench_stf2.json 2> $O/bench_stf2.err; tail -1 $O/bench_stf2.json | cut -c1-150
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench_stf2.json").read().strip().splitlines()[-1])
for k,v in d["roofline_families"].items(): print(k, v)
PY
