#!/bin/bash
# usage (GPU box, repository root): bash tools/pmc.sh <kernel-substring> <python script> [args...]
# rocprofv3 --pmc passes (one per counter group, no trace domains: MI355X_MICROARCH.md) over any probe; per group the averages per launch of
# the kernels whose name contains the substring, with the kernel name, its registers / LDS / grid.  FETCH_SIZE and WRITE_SIZE run in
# passes of their own (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, in KB, on gfx950).  Examples:
#   bash tools/pmc.sh gpfq_blk_kernel tools/pmc_probe.py 3 1 0 0 0          (the headline kernel; rounds 1-3: pmc_run.sh / prof_r0N.sh)
#   bash tools/pmc.sh gpfq_gram_s2 tools/conv1_probe.py                      (ResNet50 conv1; round 3: pmc_conv1.sh)
#   bash tools/pmc.sh gpfq_gram_shift_nhwc tools/conv3x3_probe.py 4096 56 56 64 64      (round 3: pmc_nhwc.sh)
# Round 6 added the groups VERDICT r05 asked for (instruction fetch / instruction cache, issue cycles of the memory instruction classes and
# their in-flight levels, LDS address conflicts / unaligned stalls / FIFO-full cycles and the load / store split, the scalar data cache).
#   PMC_LAYER=1 bash tools/pmc.sh gpfq_blk_kernel tools/pmc_probe.py 3 1 0 0 0        (the headline kernel as bench.py's step launches it)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
KS=$1; shift
OUT=$ROOT/gpurun_out/pmc_$KS
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH" \
           "SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES SQ_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM_NORM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=$OUT/p$i
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $ROOT/$1 "${@:2}" > $d.log 2>&1
  echo "== $set"
  python3 - "$d" "$KS" <<'PY'
import csv, glob, sys
d, ks = sys.argv[1:3]
fs = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
if not fs:
    print("  no counter file (a counter of this group is not available on this chip?)"); sys.exit(0)
rows = [r for r in csv.DictReader(open(fs[0])) if ks in r["Kernel_Name"]]
acc, meta = {}, {}
for r in rows:
    acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for name in meta:
    print("  kernel:", name[:150]); print("    vgpr/agpr/lds/workgroup/grid", meta[name])
    for (n, c), v in acc.items():
        if n == name:
            print(f"    {c:28s} {sum(v)/len(v):.6g}   (launches {len(v)})")
PY
done
