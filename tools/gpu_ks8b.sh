for cfg in "ICM_CONV_KS8=0" "ICM_CONV_KS8=1" "ICM_CONV_KS8=1 ICM_CONV_KS8_MINWG=128" "ICM_CONV_KS8=1 ICM_CONV_KS8_MINWG=192" "ICM_CONV_KS8=0" "ICM_CONV_KS8=1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  f=$(env $cfg timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  w=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v  fwd $f  stf $w"
done
