#!/bin/bash
# usage (GPU box, repository root): bash tools/pmc_nhwc.sh "n H W cin cout"
# SQ counters and HBM bytes (FETCH_SIZE, a pass of its own) of the NHWC shift kernel of one 3x3 layer, per launch.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
OUT=$ROOT/gpurun_out/pmc_nhwc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/nhwc_probe.py $1 2 > $OUT/p$i.log 2>&1
  python3 - $OUT/p$i <<'PY'
import csv,glob,sys
fs=glob.glob(sys.argv[1]+"/*/*_counter_collection.csv")+glob.glob(sys.argv[1]+"/*_counter_collection.csv")
if not fs: print("no counter file"); sys.exit(0)
acc={}
for r in csv.DictReader(open(fs[0])):
    if "shift_nhwc" in r["Kernel_Name"] and "combine" not in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"{k:26s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
  rm -rf $OUT/p$i
done
