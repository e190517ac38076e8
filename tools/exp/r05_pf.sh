#!/bin/bash
# operand rows requested 2 / 4 pair-steps ahead in the four-group narrow shapes with eight sweep wavefronts (diagnostic builds -DGPFQ_BLK_PF_NARROW=2 / 4) against one
mkdir -p gpurun_out/r05
L=gpurun_out/r05/pf.log
: > $L
{
for sh in "4096 512 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" "4096 2048 1024 1.585 3 16" "4096 128 1024 1.585 3 16" "4096 512 1000 4 5 16" "4096 1500 1000 4 5 16" "784 128 1000 4 5 16"; do
  echo "== shape $sh"
  for fl in "" "-DGPFQ_BLK_PF_NARROW=2" "-DGPFQ_BLK_PF_NARROW=4"; do
    export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
    echo -n "  [$fl] "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
cat $L
