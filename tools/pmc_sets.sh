#!/bin/bash
# usage (GPU box): tools/pmc_sets.sh <kernel-substring> "<counter set 1>;<counter set 2>;..." <python script> [args...]
# One rocprofv3 --pmc pass per counter set (no trace domains), counters of the kernels whose name contains the substring averaged per launch.
export PYTHONPATH=$GRAFT_REPO_ROOT
KS=$1; SETS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra arr <<< "$SETS"
i=0
for set in "${arr[@]}"; do
  i=$((i+1)); d=$GRAFT_REPO_ROOT/gpurun_out/pmcs_$i; rm -rf $d
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/$1 "${@:2}" > $d.log 2>&1
  python3 - "$d" "$KS" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*_counter_collection.csv")
if not fs:
    print("no counter file for this set (unknown counter?)"); sys.exit(0)
acc = {}
for r in csv.DictReader(open(fs[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"][:56], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:56s} {c:30s} {sum(v)/len(v):.5g}  (launches {len(v)})")
PY
done
