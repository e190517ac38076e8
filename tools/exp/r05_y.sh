#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/y.log
: > $L
{
echo "### parity"
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "quad or boundaries or role_split" 2>&1 | tail -3
echo "### deep form on / off"
for sh in "4096 512 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" "4096 2048 1024 1.585 3 16" "4096 128 1024 1.585 3 16" "4096 512 1000 4 5 16"; do
  echo "== shape $sh"
  for dp in 0 1; do
    echo -n "  BLK_DEEP=$dp "; BLK_DEEP=$dp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
export GPFQ_DIAG="-DGPFQ_BLK_STAMPS"
for dp in 0 1; do
  echo "### BLK_DEEP=$dp (stamps build)"
  BLK_DEEP=$dp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 512 1024 1.585 3 0 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|slot top|rror" | cut -c1-300
done
} >> $L 2>&1
cat $L
