#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/e.log
: > $L
{
echo "### variants: '' = hoist + out-of-line slow path + early header prefetch; see the flags"
for sh in "4096 4096 1024 1.585 3 0" "4096 512 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 1000 2048 4 5 0"; do
  echo "== shape $sh"
  for rep in 1 2; do
  for fl in "" "-DGPFQ_BLK_NO_HOIST" "-DGPFQ_BLK_NO_OOL" "-DGPFQ_BLK_NO_HOIST -DGPFQ_BLK_NO_OOL" "-DGPFQ_BLK_NO_HOIST -DGPFQ_BLK_NO_OOL -DGPFQ_BLK_LATE_HDR"; do
    export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
    echo -n "  [$fl] "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done; done
done
unset GPFQ_DIAG
echo "### e2e resnet50 4096 (BN statistics calibrated)"
timeout 1500 python tools/e2e_resnet50.py 4096 16 --per-layer 2>&1 | grep -v amdgpu.ids | tail -8
} >> $L 2>&1
tail -70 $L
