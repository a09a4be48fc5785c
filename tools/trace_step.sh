# kernel trace of one forward-only and one train step (cnn), CSV pulled back for per-shape analysis (tools/analyze_trace.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in fwd train; do
  extra=""; [ $mode = fwd ] && extra="--fwd-only"
  rm -rf $R/gpurun_out/trace_$mode
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$mode -- python3 $R/bench.py $extra --model ${MODEL:-cnn} --steps 1 --warmup 2 --no-cpu-baseline > $R/gpurun_out/trace_$mode.log 2>&1 || echo "trace $mode failed"
  f=$(find $R/gpurun_out/trace_$mode -name '*kernel_trace.csv' | head -1)
  # keep only the columns needed, gzip to stay small
  python3 - "$f" "$R/gpurun_out/trace_${MODEL:-cnn}_$mode.csv.gz" <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X", "LDS_Block_Size"]
with gzip.open(sys.argv[2], "wt") as f:
    w = csv.writer(f); w.writerow(keep)
    for r in rows: w.writerow([r.get(k, "") for k in keep])
print("rows", len(rows), "cols", list(rows[0].keys()))
PY
  rm -rf $R/gpurun_out/trace_$mode
done
