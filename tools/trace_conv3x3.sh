#!/bin/bash
# usage (GPU box, repository root):  bash tools/trace_conv3x3.sh ["n H W cin cout" ...]
# rocprofv3 kernel trace of ResNet50's four 3x3 layer shapes through the layer driver (tools/conv3x3_probe.py).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/conv3x3
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "4096 56 56 64 64" "4096 28 28 128 128" "4096 14 14 256 256" "4096 7 7 512 512"; fi
for sh in "$@"; do
  tag=$(echo $sh | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o t -- python3 $ROOT/tools/conv3x3_probe.py $sh > $OUT/$tag.log 2>&1
  echo "== $sh"; grep "conv_shift=1" $OUT/$tag.log
  find $OUT/$tag -name "t_kernel_stats.csv" -exec cp {} $OUT/${tag}_kernel_stats.csv \;
  grep -E "gpfq" $OUT/${tag}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150 | head -8
done
