for cfg in "ICM_MAT_MIN_PIXELS=0" "ICM_MAT_MIN_PIXELS=2048" "ICM_MATERIALIZE=0" "ICM_MAT_MIN_PIXELS=2048 ICM_CONV_DMA=0"; do
  v=$(env $cfg timeout -k 10 300 python bench.py --model stf6 --no-cpu-baseline --no-shape-table --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> stf6 train $v"
done
for cfg in "ICM_MAT_MIN_PIXELS=0" "ICM_MAT_MIN_PIXELS=2048"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  w=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v stf $w"
done
