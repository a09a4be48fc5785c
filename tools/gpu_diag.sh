set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2; mkdir -p $O
for dg in 0 1 2 3; do
  echo "== ICM_WG_DIAG=$dg"
  ICM_WG_DIAG=$dg timeout -k 10 200 python tools/tune_wgrad.py "1x1 192->192" 2>&1 | grep -v amdgpu | cut -c1-160
done
cd /tmp && export TMPDIR=/tmp
for dg in 0 1 2; do
  rm -rf $O/prof_d$dg
  ICM_WG_DIAG=$dg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d$dg -- python3 $R/tools/tune_wgrad.py "192->192 @64" > /dev/null 2>&1
  f=$(find $O/prof_d$dg -name '*kernel_stats.csv' | head -1); echo "== diag $dg"; grep -E "t33_kernel<6, 6|reduce" "$f" | cut -c1-120; rm -rf $O/prof_d$dg
done
