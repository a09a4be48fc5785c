set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
run() { timeout -k 10 "$@"; rc=$?; if [ $rc -ge 124 ]; then echo "KILLED rc=$rc: $*"; exit $rc; fi; return $rc; }
ICM_WG_DMA=2 run 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wacnn.py -q -p no:cacheprovider --timeout 600 -k "wgrad or oracle_small or trainer_two or gdn" > $O/t_dma2.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" $O/t_dma2.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/t_dma2.log | head -20
for dm in 1 2; do echo "== ICM_WG_DMA=$dm"; ICM_WG_DMA=$dm run 300 python tools/tune_wgrad.py "x" 2>&1 | grep -E "5x5|gelu" | cut -c1-160; done
for dm in 1 2; do ICM_WG_DMA=$dm run 300 python bench.py --no-cpu-baseline --no-shape-table > $O/bench9_$dm.json 2> $O/bench9.err; tail -1 $O/bench9_$dm.json | cut -c1-150; done
