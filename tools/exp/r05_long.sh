#!/bin/bash
# cluster form beyond 16384 samples (up to GPFQ_ONCHIP_MAX_M = 28672: 28 slices, gathered in batches of eight)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/long.log
: > $L
{
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -3
for sh in "4096 4096 20000 1.585 3 16" "4096 1024 28672 1.585 3 16" "4096 4096 16384 1.585 3 8"; do
  echo "== shape $sh"
  PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|rror|!!" | cut -c1-200
done
} >> $L 2>&1
cat $L
