cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-11}; CFG=${2:-7}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc2_$tag -- python3 $R/tools/one_conv.py $SH $CFG wgrad > $R/gpurun_out/pmc2_$tag.log 2>&1 || echo "pmc $tag failed"
done
