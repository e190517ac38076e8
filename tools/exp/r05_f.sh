#!/bin/bash
# round 5, GPU call F: the whole -m gpu suite, the bench line, the round's profile set, the configs + shapes table (perf guard baseline), fuzz on HEAD
mkdir -p gpurun_out/r05
L=gpurun_out/r05/f.log
: > $L
{
echo "### pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "### bench.py"
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "### bench_configs --resnet --shapes"
timeout 1800 python tools/bench_configs.py --resnet --shapes > gpurun_out/r05/configs.log 2>&1; tail -3 gpurun_out/r05/configs.log | cut -c1-300
cp gpurun_out/configs.json gpurun_out/r05/configs.json
echo "### fuzz 240 s"
timeout 600 python tools/fuzz_parity.py 240 505 2>&1 | tail -6
echo "### prof_round r05"
ROUND=r05 timeout 2400 bash tools/prof_round.sh > gpurun_out/r05/prof_round.log 2>&1; tail -5 gpurun_out/r05/prof_round.log | cut -c1-300
} >> $L 2>&1
tail -40 $L
