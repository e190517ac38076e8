#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/x.log
: > $L
{
export GPFQ_DIAG="-DGPFQ_BLK_STAMPS"
for dp in 0 1; do
  echo "### BLK_DEEP=$dp (stamps build)"
  BLK_DEEP=$dp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 512 1024 1.585 3 0 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|slot top|rror" | cut -c1-300
done
} >> $L 2>&1
cat $L
