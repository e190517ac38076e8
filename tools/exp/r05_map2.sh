#!/bin/bash
# workgroup map chosen by rounds: slice counts that do not divide 8
mkdir -p gpurun_out/r05
L=gpurun_out/r05/map2.log
: > $L
{
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -3
for sh in "4096 4096 20000 1.585 3 8" "4096 2048 20000 1.585 3 8" "4096 1024 28672 1.585 3 8" "4096 3000 6000 4 5 8" "4096 2000 5008 3 4 8" "4096 4096 12000 1.585 3 8"; do
  echo "== shape $sh"
  for mp in 0 1 -1; do
    echo -n "  map $mp "; BLK_CLUSTER_MAP=$mp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
cat $L
