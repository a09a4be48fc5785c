set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "ks8" 2>&1 | tail -8 || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wacnn.py tests/test_gpu_b16.py -x -q 2>&1 | tail -4 || exit 1
for cfg in "ICM_CONV_KS8=0" "ICM_CONV_KS8=1" "ICM_CONV_KS8=1 ICM_CONV_KS8_MAXWG=384" "ICM_CONV_KS8=1 ICM_CONV_KS8_MAXWG=1536" "ICM_CONV_KS8=0" "ICM_CONV_KS8=1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  f=$(env $cfg timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v  fwd $f"
done
ICM_SHAPE_TABLE=$O/shapes_ks8.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_ks8.json 2>$O/bench_ks8.err
python -c "import json; r=json.loads(open('$O/bench_ks8.json').read().strip().splitlines()[-1]); print('train', r['value'], {k:(v['ms'],v['tflops']) for k,v in r['roofline_families'].items()})"
