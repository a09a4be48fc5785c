set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or bench_size" 2>&1 | tail -6 || exit 1
for cfg in "ICM_WG_TAP9=0" "ICM_WG_TAP9=1" "ICM_WG_TAP9=0" "ICM_WG_TAP9=1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  w=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v  stf $w"
done
ICM_SHAPE_TABLE=$O/shapes_tap9.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_tap9.json 2>$O/bench_tap9.err
python -c "import json; r=json.loads(open('$O/bench_tap9.json').read().strip().splitlines()[-1]); print('train', r['value'], {k:(v['ms'],v['tflops']) for k,v in r['roofline_families'].items()})"
