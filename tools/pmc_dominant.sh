# rocprofv3 PMC passes for the dominant kernel (g_a.2 forward conv, auto tile config); separate passes per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcd_*
for set in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcd_$tag -- python3 $R/tools/one_conv.py 0 -1 > $R/gpurun_out/pmcd_$tag.log 2>&1 || echo "pmc $tag failed"
done
