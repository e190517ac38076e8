#!/bin/bash
# rocprofv3 profile of the cluster form: 4096 x 4096 on 8192 samples, ternary (tools/pmc_probe.py, PMC_M=8192): kernel trace, then FETCH_SIZE / WRITE_SIZE and SQ passes of their own
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r05_cluster
mkdir -p $OUT
export PYTHONPATH=$ROOT PMC_M=8192
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cl -- python3 $ROOT/tools/pmc_probe.py 3 1 0 0 0 > $OUT/trace.log 2>&1
find $OUT/trace -name "cl_kernel_stats.csv" -exec cp {} $OUT/cluster_8192_kernel_stats.csv \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); d=$OUT/pmc$i
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $d -- python3 $ROOT/tools/pmc_probe.py 3 1 0 0 0 > $d.log 2>&1)
  echo "== $set" >> $OUT/cluster_8192_counters.txt
  python3 - "$d" >> $OUT/cluster_8192_counters.txt <<'PY'
import csv, glob, sys
d = sys.argv[1]
fs = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
if not fs:
    print("no counter file"); sys.exit(0)
rows = [r for r in csv.DictReader(open(fs[0])) if "gpfq_blk" in r["Kernel_Name"]]
acc = {}
for r in rows:
    acc.setdefault((r["Kernel_Name"][:110], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    print(f"  {n}  {c:24s} {sum(v)/len(v):.6g}  (launches {len(v)}; grid {rows[0]['Grid_Size']}, workgroup {rows[0]['Workgroup_Size']}, vgpr {rows[0]['VGPR_Count']}, lds {rows[0]['LDS_Block_Size']})")
PY
done
cd $ROOT
head -8 $OUT/cluster_8192_kernel_stats.csv | cut -c1-220
cat $OUT/cluster_8192_counters.txt | cut -c1-330
