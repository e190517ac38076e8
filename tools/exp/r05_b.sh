#!/bin/bash
# round 5, GPU call B: f32-MFMA micro-benchmark, graph capture on the GPU, sharded capture after the grid change, conv1 on signed images, cfg5 end to end
mkdir -p gpurun_out/r05
L=gpurun_out/r05/b.log
: > $L
{
echo "### ubench mfma_f32_chain"
timeout 300 tools/ubench/mfma_f32_chain
echo "### functional + multirank tests"
timeout 1500 python -m pytest tests/test_functional_gpu.py tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -8
echo "### conv1 signed"
timeout 1200 python -m pytest tests/test_fullsize_configs.py -m gpu -x -q -s -k "conv1_signed" 2>&1 | tail -12
echo "### e2e resnet50"
timeout 1500 python tools/e2e_resnet50.py 4096 16 --reference-capture 64 2>&1 | grep -v amdgpu.ids | tail -8
} >> $L 2>&1
tail -70 $L
