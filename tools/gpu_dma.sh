set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --timeout 600 -k "wgrad or conv_fwd_bwd or bench_size or attention" > $O/t_dma.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_dma.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_dma.log | head -20
for dm in 1 0; do echo "== ICM_WG_DMA=$dm"; ICM_WG_DMA=$dm run 300 python tools/tune_wgrad.py "x" 2>&1 | grep -E "5x5|3x3 480|3x3 224|3x3 96" | cut -c1-160; done
run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench8.json 2> $O/bench8.err; tail -1 $O/bench8.json | cut -c1-150
