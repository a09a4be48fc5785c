# rocprofv3 PMC passes for one conv shape (tools/tune_conv.py SHAPES index $1, config $2 or -1 = auto)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
IDX=${1:-11}; CFG=${2:--1}
rm -rf $R/gpurun_out/pmcs_*
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcs_$tag -- python3 $R/tools/one_conv.py $IDX $CFG > $R/gpurun_out/pmcs_$tag.log 2>&1 || echo "pmc $tag failed"
  f=$(find $R/gpurun_out/pmcs_$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    if "conv_igemm" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
