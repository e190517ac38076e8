#!/bin/bash
# cluster form of the block kernel: first run -- parity tests, then timings against the wide kernel
mkdir -p gpurun_out/r05
L=gpurun_out/r05/p.log
: > $L
{
echo "### cluster form: parity"
timeout 900 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -15
echo "### timings (pipe_probe: old kernel = pipe 0, then the block kernel as dispatched)"
for sh in "4096 4096 8192 1.585 3 64" "4096 4096 6000 4 5 32" "4096 1024 8192 1.585 3 32"; do
  echo "== shape $sh"
  PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -v amdgpu.ids | cut -c1-230
done
echo "== shape 4096 4096 5008 3 4 32: cluster threshold 4096 against the classic shape"
for th in 1 4096; do
  echo "-- BLK_CLUSTER=$th"
  BLK_CLUSTER=$th PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 4096 5008 3 4 32 2>&1 | grep -E "pipe mode|old kernel|rror|!!" | cut -c1-230
done
} >> $L 2>&1
tail -60 $L
