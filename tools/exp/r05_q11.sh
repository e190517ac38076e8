#!/bin/bash
# four-group one-neuron-per-lane shapes on rows of 769..1024 samples: eleven sweep wavefronts of three pairs (diagnostic build -DGPFQ_BLK_QUAD11) against eight
mkdir -p gpurun_out/r05
L=gpurun_out/r05/q11.log
: > $L
{
export GPFQ_DIAG="-DGPFQ_BLK_QUAD11"
for sh in "4096 512 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" "4096 128 1024 1.585 3 16" "4096 512 1000 4 5 16" "784 128 1000 4 5 16"; do
  echo "== shape $sh"
  for nw in 8 11; do
    echo -n "  BLK_QUAD_NW=$nw "; BLK_QUAD_NW=$nw PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
export GPFQ_DIAG="-DGPFQ_BLK_QUAD11 -DGPFQ_BLK_STAMPS"
for nw in 8 11; do
  echo "### BLK_QUAD_NW=$nw (stamps build)"
  BLK_QUAD_NW=$nw PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 512 1024 1.585 3 0 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|slot top|rror" | cut -c1-300
done
} >> $L 2>&1
cat $L
