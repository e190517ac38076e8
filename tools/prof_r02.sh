#!/bin/bash
# usage (GPU box, from the repository root):  bash tools/prof_r02.sh
# Round-2 profile set of the headline benchmark: rocprofv3 kernel trace + stats of `python3 bench.py`, then separate
# --pmc passes (no trace domains) over tools/pmc_probe.py for the dominant kernel's SQ counters and its HBM traffic
# (FETCH_SIZE / WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes).  Everything lands in gpurun_out/prof_r02.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $ROOT/bench.py > $OUT/bench.log 2>&1
cd $ROOT
KS=${KS:-gpfq_blk_kernel}
summ() {   # average of every counter of the kernels whose name contains $KS
python3 - "$1" "$KS" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*_counter_collection.csv")
if not fs:
    print("no counter file in", sys.argv[1]); sys.exit(0)
acc, meta = {}, None
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):.6g}   (launches {len(v)})")
print("vgpr/agpr/lds/workgroup/grid", meta)
PY
}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=$OUT/pmc$i
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $d -- python3 $ROOT/tools/pmc_probe.py 3 1 0 0 0 > $d.log 2>&1)
  echo "== $set" >> $OUT/counters.txt
  summ $d >> $OUT/counters.txt
done
cp $OUT/bench/*/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || cp $OUT/bench/bench_kernel_stats.csv $OUT/ 2>/dev/null
tail -3 $OUT/bench.log; cat $OUT/counters.txt; head -4 $OUT/bench_kernel_stats.csv | cut -c1-220
