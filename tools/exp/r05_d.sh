#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/d.log
: > $L
{
echo "### hoist A/B (shipped = out-of-line slow path + first requests behind the control word; NO_HOIST = round 4's order)"
for sh in "4096 4096 1024 1.585 3 0" "4096 512 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 4096 768 1.585 3 0"; do
  SHAPE="$sh" bash tools/blk_ab.sh flags "" "-DGPFQ_BLK_NO_HOIST" "" "-DGPFQ_BLK_NO_HOIST" 2>&1 | grep -E "flags|pipe mode"
done
echo "### parity (role split / block kernel / slow path)"
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -5
echo "### multirank"
timeout 1500 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -8
echo "### e2e resnet50 4096"
timeout 1500 python tools/e2e_resnet50.py 4096 16 --per-layer 2>&1 | grep -v amdgpu.ids | tail -8
} >> $L 2>&1
tail -60 $L
