#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/m.log
: > $L
{
echo "### seven sweep wavefronts with operand rows requested two pair-steps ahead: blk_quad_waves 7 / 8, rows of 769..1024 samples too"
for sh in "4096 512 1024 1.585 3 0" "4096 1024 1024 1.585 3 0" "4096 128 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 512 1000 4 5 0" "4096 1024 768 1.585 3 0" "4096 1024 512 1.585 3 0" "784 128 512 4 5 0" "4096 2048 768 1.585 3 0"; do
  echo "== shape $sh"
  for rep in 1 2; do for nw in 7 8; do
    echo -n "  [blk_quad_waves $nw] "; BLK_QUAD_NW=$nw PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done; done
done
echo "### parity (all blk tests) and fuzz"
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
timeout 600 python tools/fuzz_parity.py 100 1212 2>&1 | tail -2
} >> $L 2>&1
tail -60 $L
