#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/o.log
: > $L
{
echo "### outputs written by the sweep wavefronts (block b - 1 at the top of slot b), no flush in the decision wavefront"
for sh in "4096 512 1024 1.585 3 0" "4096 1024 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 4096 1024 1.585 3 0" "4096 1024 768 1.585 3 0" "4096 512 1000 4 5 0" "4096 1000 2048 4 5 0" "2048 128 5008 3 4 0" "784 128 512 4 5 0" "4096 4096 2048 4 5 0" "4096 4096 768 1.585 3 0"; do
  echo "== shape $sh"
  for rep in 1 2; do
    echo -n "  "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done
done
echo "### parity"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fullsize_configs.py -m gpu -x -q -k "not conv" 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "### fuzz 150 s"
timeout 600 python tools/fuzz_parity.py 150 1313 2>&1 | tail -2
echo "### bench"
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500
} >> $L 2>&1
tail -50 $L
