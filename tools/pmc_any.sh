#!/bin/bash
# usage: tools/pmc_any.sh <kernel-substring> <python script> [args...]   (run on the GPU box)
# SQ / cache counters of the kernels whose name contains the substring, averaged per launch (separate passes).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KS=$1; shift
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MUL_F64 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  d=gpurun_out/pmc_$(echo $set $@ | md5sum | cut -c1-6)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $@ > /dev/null 2>&1
  python3 - "$d" "$KS" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
if not fs:
    print("no counter file for this set"); sys.exit(0)
acc = {}
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:48s} {c:28s} {sum(v)/len(v):.4g}")
PY
done
