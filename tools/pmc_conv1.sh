#!/bin/bash
# usage (GPU box, repository root): bash tools/pmc_conv1.sh -- SQ counters of ResNet50's conv1 (gpfq_gram_s2_kernel), one --pmc pass per group
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/pmc_conv1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/conv1_probe.py > $OUT/p$i.log 2>&1
  python3 - $OUT/p$i <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+"/*/*_counter_collection.csv")+glob.glob(sys.argv[1]+"/*_counter_collection.csv")
acc={}
for r in csv.DictReader(open(fs[0])):
    if "gram_s2_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:26s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
done
