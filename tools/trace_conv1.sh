#!/bin/bash
# usage (GPU box, repository root):  bash tools/trace_conv1.sh
# rocprofv3 kernel trace of the ResNet50 conv1 layer driver (tools/conv1_probe.py), distinct tensors and first-layer form.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/conv1
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -o conv1 -- python3 $ROOT/tools/conv1_probe.py > $OUT/conv1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f -o conv1_first -- python3 $ROOT/tools/conv1_probe.py --first > $OUT/conv1_first.log 2>&1
cd $ROOT
find $OUT/d -name "conv1_kernel_stats.csv" -exec cp {} $OUT/conv1_kernel_stats.csv \;
find $OUT/f -name "conv1_first_kernel_stats.csv" -exec cp {} $OUT/conv1_first_kernel_stats.csv \;
head -12 $OUT/conv1_kernel_stats.csv | cut -c1-150
head -8 $OUT/conv1_first_kernel_stats.csv | cut -c1-150
