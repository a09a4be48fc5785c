set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
T=${1:-x}
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -k "conv" 2>&1 | tail -5 || exit 1
ICM_SHAPE_TABLE=$O/shapes_$T.jsonl timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$T.json 2>$O/bench_$T.err || exit 1
python -c "import json; r=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('train', r['value'], r['roofline_families'])"
timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --no-shape-table | tail -1 | python -c "import sys,json; print('fwd', round(json.loads(sys.stdin.read())['value'],1))"
timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table | tail -1 | python -c "import sys,json; print('stf', round(json.loads(sys.stdin.read())['value'],1))"
