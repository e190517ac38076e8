#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/c.log
: > $L
{
echo "### functional tests"
timeout 900 python -m pytest tests/test_functional_gpu.py -m gpu -x -q 2>&1 | tail -8
echo "### e2e resnet50 512 images, per layer + profile"
timeout 1500 python tools/e2e_resnet50.py 512 16 --per-layer --profile 2>&1 | grep -v amdgpu.ids | tail -75
echo "### multirank tests"
timeout 1500 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -8
} >> $L 2>&1
tail -120 $L
