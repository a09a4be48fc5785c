for cfg in "A=1" "ICM_HOLD_CHAIN_WGRADS=1" "ICM_WG_EVERY=16 ICM_WG_MIN=16" "ICM_WG_EVERY=-1" "A=1" "ICM_HOLD_CHAIN_WGRADS=1" "ICM_PACK_WINDOW=48" "ICM_CONV_BOOST=1.4" "ICM_CONV_BOOST=1.1"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> $v img/s"
done
