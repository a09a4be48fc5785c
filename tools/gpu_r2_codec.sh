# codec tests + full per-shape table of the train step and the forward
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python -m pytest tests/test_gpu_codec.py -q -p no:cacheprovider -rA --timeout 600 > $O/t_codec.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_codec.log | tail -2
grep -E "^(FAILED|ERROR)" $O/t_codec.log | head -30
ICM_SHAPE_TABLE=$O/shapes_train.jsonl run 600 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"
ICM_SHAPE_TABLE=$O/shapes_fwd.jsonl run 300 python bench.py --fwd-only --no-cpu-baseline > $O/bench2_fwd.json 2>> $O/bench2.err
tail -1 $O/bench2_fwd.json | cut -c1-200
