#!/bin/bash
# cluster form: workgroup maps (a cluster inside one XCD / consecutive ids) by slice count
mkdir -p gpurun_out/r05
L=gpurun_out/r05/t.log
: > $L
{
echo "### cluster form: parity"
timeout 900 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -8
for sh in "4096 4096 8192 1.585 3 64" "4096 4096 7000 1.585 3 32" "4096 4096 6000 4 5 32" "4096 4096 5008 3 4 32" "4096 4096 4096 4 5 32" "4096 4096 3000 4 5 32" "4096 4096 2048 4 5 32" "4096 2048 3000 4 5 32" "4096 1000 3000 4 5 32" "4096 1000 4096 4 5 32"; do
  echo "== shape $sh"
  echo -n "  classic / default      "; BLK_CLUSTER=1 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  for mp in 0 1; do
    echo -n "  cluster from 1024, map $mp "; BLK_CLUSTER_MAP=$mp BLK_CLUSTER=1024 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
tail -70 $L
