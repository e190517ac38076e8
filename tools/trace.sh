#!/bin/bash
# usage (GPU box, repository root): bash tools/trace.sh <tag> <python script> [args...]
# rocprofv3 --kernel-trace --stats of any probe; the per-kernel summary lands in gpurun_out/trace/<tag>_kernel_stats.csv (copied to profiles/r0N/
# when it is evidence) and its first rows are printed.  Rounds 1-3 had one script per probe (trace_conv1.sh, trace_conv3x3.sh, mlp_prof.sh):
#   bash tools/trace.sh conv1 tools/conv1_probe.py            bash tools/trace.sh conv1_first tools/conv1_probe.py --first
#   bash tools/trace.sh conv3x3_4096_56_56_64_64 tools/conv3x3_probe.py 4096 56 56 64 64 --host-inputs --shift-only
#   bash tools/trace.sh mlp tools/e2e_mlp.py 25000 500,300 1.585
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$ROOT
TAG=$1; shift
OUT=$ROOT/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -o t -- python3 $ROOT/$1 "${@:2}" > $OUT/$TAG.log 2>&1
find $OUT/$TAG -name "t_kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
tail -4 $OUT/$TAG.log | cut -c1-200
head -14 $OUT/${TAG}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
