#!/bin/bash
# cluster form with D two slots old (the exchange off the chain of decisions)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/s.log
: > $L
{
echo "### cluster form: parity"
timeout 900 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -8
for sh in "4096 4096 8192 1.585 3 64" "4096 4096 6000 4 5 32" "4096 4096 5008 3 4 32" "4096 4096 4096 4 5 32" "4096 4096 3000 4 5 32" "4096 4096 2048 4 5 32" "4096 4096 1536 4 5 32" "4096 1000 2048 4 5 32" "4096 2048 2048 4 5 32"; do
  echo "== shape $sh"
  for th in 1 1024; do
    echo -n "  BLK_CLUSTER=$th "; BLK_CLUSTER=$th PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-200
  done
done
} >> $L 2>&1
tail -60 $L
