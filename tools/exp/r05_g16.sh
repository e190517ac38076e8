#!/bin/bash
# gather batches of sixteen slices for clusters of nine and more
mkdir -p gpurun_out/r05
L=gpurun_out/r05/g16.log
: > $L
{
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -3
for sh in "4096 4096 8192 1.585 3 8" "4096 4096 16384 1.585 3 8" "4096 4096 20000 1.585 3 8" "4096 4096 12000 1.585 3 8" "2048 128 5008 3 4 8" "4096 4096 5008 3 4 8" "4096 1024 28672 1.585 3 8"; do
  echo "== shape $sh"
  PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
done
} >> $L 2>&1
cat $L
