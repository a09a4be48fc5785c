# rocprofv3 --kernel-trace --stats of the bench with the weight gradients on the main stream (ICM_WG_EVERY=-1): kernels
# never overlap, so the per-kernel durations are comparable with bench.py's serialised family table
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ICM_WG_EVERY=-1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --no-cpu-baseline --no-shape-table > $O/prof_serial.log 2>&1 || { tail -5 $O/prof_serial.log; exit 1; }
grep '"metric"' $O/prof_serial.log | tail -1 | cut -c1-200
f=$(find $O/prof_serial -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_serial.csv; rm -rf $O/prof_serial
python3 $R/tools/kernel_stats_families.py $O/kernel_stats_serial.csv 13
