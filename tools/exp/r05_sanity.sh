#!/bin/bash
# quick check of the shipped build after the reverted experiments: cluster tests, dense parity, bench line
mkdir -p gpurun_out/r05
L=gpurun_out/r05/sanity.log
: > $L
{
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900
PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 2048 128 5008 3 4 16 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-200
PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 4096 8192 1.585 3 16 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-200
} >> $L 2>&1
cat $L
