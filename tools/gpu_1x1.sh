set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "conv1x1 or conv_fwd_bwd or gate or gdn" 2>&1 | tail -15 || exit 1
for cfg in "ICM_CONV_1X1=0" "ICM_CONV_1X1=-1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg train -> $v img/s"
  v=$(env $cfg timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg fwd -> $v img/s"
  v=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg stf train -> $v img/s"
done
ICM_SHAPE_TABLE=$O/shapes_1x1.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_1x1.json 2>$O/bench_1x1.err
tail -c 600 $O/bench_1x1.json
