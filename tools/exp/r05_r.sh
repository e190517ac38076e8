#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/r.log
: > $L
{
for fl in "-DGPFQ_CL_NEAR_INV" ""; do
  export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
  echo "### build [$fl]"
  for sh in "4096 4096 8192 1.585 3 64" "4096 4096 5008 3 4 32"; do
    echo "== shape $sh (cluster threshold 4096)"
    BLK_CLUSTER=4096 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-230
  done
done
} >> $L 2>&1
tail -60 $L
