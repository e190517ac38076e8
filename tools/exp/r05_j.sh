#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/j.log
: > $L
{
echo "### narrow shapes with SEVEN sweep wavefronts (two wavefronts per SIMD): '' = eight (shipped), QUAD7, QUAD7 + headers behind the chain"
for sh in "4096 512 1024 1.585 3 0" "4096 1024 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 128 1024 1.585 3 0" "784 128 512 4 5 0" "4096 1024 768 1.585 3 0" "4096 1024 512 1.585 3 0" "4096 512 1000 4 5 0"; do
  echo "== shape $sh"
  for rep in 1 2; do
  for fl in "" "-DGPFQ_BLK_QUAD7" "-DGPFQ_BLK_QUAD7 -DGPFQ_BLK_HDR_CHAIN"; do
    export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
    echo -n "  [$fl] "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done; done
done
export GPFQ_DIAG="-DGPFQ_BLK_QUAD7 -DGPFQ_BLK_HDR_CHAIN"
echo "### parity with QUAD7 + HDR_CHAIN"
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -4
echo "### fuzz 100 s with QUAD7 + HDR_CHAIN"
timeout 600 python tools/fuzz_parity.py 100 909 2>&1 | tail -3
} >> $L 2>&1
tail -70 $L
