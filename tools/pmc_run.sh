#!/bin/bash
# usage: tools/pmc_run.sh <kernel-substring> <pmc_probe args...>   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KS=$1; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  d=gpurun_out/pmc_$(echo $set $@ | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 tools/pmc_probe.py $@ > /dev/null 2>&1
  python3 - "$d" "$KS" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = {}
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        vg = r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"]
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):.4g}")
print("vgpr/agpr/lds/wg", vg)
PY
done
