#!/bin/bash
# cluster form against the classic shapes in narrow layers on long rows (BLK_CLUSTER=16384: no row takes the cluster form)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/u.log
: > $L
{
for sh in "2048 128 5008 3 4 16" "128 10 5008 3 4 8" "4096 10 5008 3 4 8" "4096 128 4096 4 5 16" "4096 512 4096 4 5 16" "4096 512 5008 3 4 16" "4096 1024 5008 3 4 16" "4096 2048 4096 4 5 16" "4096 2048 5008 3 4 16" "4096 300 3000 4 5 16" "4096 128 3000 4 5 16" "4096 1280 3000 4 5 16" "4096 1500 2048 4 5 16" "25088 4096 2048 4 5 8"; do
  echo "== shape $sh"
  for th in 16384 1; do
    echo -n "  BLK_CLUSTER=$th "; BLK_CLUSTER=$th PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
tail -70 $L
