#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/n.log
: > $L
{
echo "### what the decision wavefront's output flush costs (NOFLUSH: outputs not written, time only)"
for sh in "4096 512 1024 1.585 3 0" "4096 1024 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 4096 1024 1.585 3 0" "4096 1024 768 1.585 3 0" "4096 512 1000 4 5 0" "4096 1000 2048 4 5 0" "2048 128 5008 3 4 0"; do
  echo "== shape $sh"
  for rep in 1 2; do for fl in "" "-DGPFQ_BLK_X_NOFLUSH"; do
    export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
    echo -n "  [$fl] "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-60
  done; done
done
} >> $L 2>&1
tail -50 $L
