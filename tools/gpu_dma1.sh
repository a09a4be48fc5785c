set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad or bench_size" 2>&1 | tail -6 || exit 1
timeout -k 10 300 python tools/tune_wgrad.py 1x1 2>&1 | grep -v gelu | cut -c1-260
for cfg in "ICM_WG_DMA1=0" "ICM_WG_DMA1=1" "ICM_WG_DMA1=0" "ICM_WG_DMA1=1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  w=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v  stf $w"
done
