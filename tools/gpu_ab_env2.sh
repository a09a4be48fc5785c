set -o pipefail
for cfg in "ICM_1X1_MIN_WAVES=1024" "ICM_1X1_MIN_WAVES=512" "ICM_1X1_MIN_WAVES=256" "ICM_1X1_MIN_WAVES=128" "ICM_1X1_MIN_WAVES=1024" "ICM_1X1_MIN_WAVES=256"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  w=$(env $cfg timeout -k 10 200 python bench.py --model stf --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  f=$(env $cfg timeout -k 10 200 python bench.py --fwd-only --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> train $v  stf $w  fwd $f"
done
