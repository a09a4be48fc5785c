# rocprofv3 PMC passes (separate runs per counter set: FETCH_SIZE and WRITE_SIZE do not fit one pass) for the heaviest
# launches of the GEMM families: the 5x5 stride-2 weight gradient, a 1x1 weight gradient, the g_a.2 forward conv, and
# (round 3) the Winograd convolution / weight-gradient kernels on the slice-chain shapes.
# Output: gpurun_out/pmcf_<what>_<set>/ ; summarise with tools/pmc_family_summary.py -> profiles/r03_pmc_family.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcf_*
for what in "wg5:one_wgrad.py:5x5s2 192->192 @128" "wg1:one_wgrad.py:1x1 192->192 @64" "conv:one_conv.py:0 -1" "wino:one_wino.py:conv" "wwino:one_wino.py:wgrad"; do
  tag=${what%%:*}; rest=${what#*:}; script=${rest%%:*}; arg=${rest#*:}
  for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    st=$(echo $set | tr ' ' '_' | cut -c1-28)
    if [ "$script" = one_conv.py ] || [ "$script" = one_wino.py ]; then
      timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcf_${tag}_$st -- python3 $R/tools/$script $arg > $R/gpurun_out/pmcf_${tag}_$st.log 2>&1 || echo "pmc $tag $st failed"
    else
      timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcf_${tag}_$st -- python3 $R/tools/$script "$arg" > $R/gpurun_out/pmcf_${tag}_$st.log 2>&1 || echo "pmc $tag $st failed"
    fi
  done
done
