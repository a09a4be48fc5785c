set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
for cfg in "A=1" "ICM_HOLD_CHAIN_WGRADS=1" "ICM_WG_EVERY=16" "ICM_WG_EVERY=32 ICM_WG_MIN=16" "ICM_WG_EVERY=4 ICM_WG_MIN=4" "A=1" "ICM_HOLD_CHAIN_WGRADS=1" "ICM_PACK_WINDOW=48" "ICM_PACK_WINDOW=12"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-shape-table 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value'],1))")
  echo "$cfg -> $v img/s"
done
