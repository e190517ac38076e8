#!/bin/bash
# deep form (narrow layers on rows of 769..1024 samples) and the cluster form with 4 / 8 neurons per workgroup
mkdir -p gpurun_out/r05
L=gpurun_out/r05/w.log
: > $L
{
echo "### parity"
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "quad or boundaries or role_split" 2>&1 | tail -4
echo "### deep form on / off"
for sh in "4096 512 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" "4096 2048 1024 1.585 3 16" "4096 128 1024 1.585 3 16" "4096 512 1000 4 5 16" "4096 10 1024 1.585 3 8"; do
  echo "== shape $sh"
  for dp in 0 1; do
    echo -n "  BLK_DEEP=$dp "; BLK_DEEP=$dp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
echo "### cluster form: neurons per workgroup"
for sh in "2048 128 5008 3 4 16" "4096 512 4096 4 5 16" "4096 1024 5008 3 4 16" "4096 1000 2048 4 5 16" "4096 1000 3000 4 5 16" "4096 300 8192 1.585 3 16" "4096 2048 2048 4 5 16"; do
  echo "== shape $sh"
  for nl in 4 2 1; do
    echo -n "  BLK_CLUSTER_NL=$nl "; BLK_CLUSTER=1536 BLK_CLUSTER_NL=$nl PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
tail -80 $L
