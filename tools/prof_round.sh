#!/bin/bash
# usage (GPU box, from the repository root):  [ROUND=r04] bash tools/prof_round.sh
# A round's profile set (rounds 2-4 ran it as prof_r02.sh / prof_r03.sh / prof_r04.sh: git history).  Everything lands in
# gpurun_out/prof_$ROUND; what is committed under profiles/$ROUND/ is copied from there.
#  1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (the headline line) -> bench_kernel_stats.csv
#  2. the same for tools/bench_configs.py (cfg1, cfg3 fc2 / fc1 / predictions, cfg4 all layers) -> configs_kernel_stats.csv
#  (round 4: + the matrix-instruction counters of the headline kernel, whose dot products moved to v_mfma_f64_4x4x4_4b_f64;
#   bench_configs.py runs with --resnet so that cfg5's layers are in the trace and in configs.json)
#  3. separate --pmc passes (no trace domains) over tools/pmc_probe.py: SQ counters, then FETCH_SIZE and WRITE_SIZE in passes
#     of their own as MI355X_MICROARCH.md prescribes.  Every pass is summarised for the kernels whose name contains $KS -- the
#     name is printed with the numbers -- and the matching ROWS of the raw counter CSV are kept (pmc_<set>_<kernel>.csv).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r04}
OUT=$ROOT/gpurun_out/prof_$ROUND
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $ROOT/bench.py > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -o configs -- python3 $ROOT/tools/bench_configs.py --resnet > $OUT/configs.log 2>&1
cd $ROOT
find $OUT/bench -name "bench_kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
find $OUT/configs -name "configs_kernel_stats.csv" -exec cp {} $OUT/configs_kernel_stats.csv \;
cp gpurun_out/configs.json $OUT/configs.json 2>/dev/null
KS=${KS:-gpfq_blk_kernel}
summ() {   # $1 = pass directory, $2 = tag: per-kernel-name averages of every counter + the raw rows of the matching kernels
python3 - "$1" "$KS" "$OUT" "$2" <<'PY'
import csv, glob, sys
d, ks, out, tag = sys.argv[1:5]
fs = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
if not fs:
    print("no counter file in", d); sys.exit(0)
rows = [r for r in csv.DictReader(open(fs[0])) if ks in r["Kernel_Name"]]
if not rows:
    print("no kernel named *%s* in %s" % (ks, fs[0])); sys.exit(0)
with open("%s/pmc_%s_%s.csv" % (out, tag, ks), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
acc = {}
for r in rows:
    acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
meta = {r["Kernel_Name"]: (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"]) for r in rows}
for name in meta:
    print("kernel:", name[:160])
    print("  vgpr/agpr/lds/workgroup/grid", meta[name])
    for (n, c), v in acc.items():
        if n == name:
            print(f"  {c:28s} {sum(v)/len(v):.6g}   (launches {len(v)})")
PY
}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=$OUT/pmc$i
  tag=$(echo $set | tr ' ' '\n' | head -1 | tr 'A-Z' 'a-z')
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $d -- python3 $ROOT/tools/pmc_probe.py 3 1 0 0 0 > $d.log 2>&1)
  echo "== $set" >> $OUT/counters.txt
  summ $d $tag >> $OUT/counters.txt
done
tail -2 $OUT/bench.log | cut -c1-600; cat $OUT/counters.txt; head -6 $OUT/bench_kernel_stats.csv | cut -c1-220; head -12 $OUT/configs_kernel_stats.csv | cut -c1-200
