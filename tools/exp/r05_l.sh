#!/bin/bash
# round 5, GPU call L: final validation of HEAD: the whole suite, bench, the profile set, configs + shapes (new perf-guard reference), fuzz
mkdir -p gpurun_out/r05
L=gpurun_out/r05/l.log
: > $L
{
echo "### pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
echo "### bench.py"
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "### bench_configs --resnet --shapes --check profiles/r05/configs.json"
timeout 2400 python tools/bench_configs.py --resnet --shapes --check profiles/r05/configs.json > gpurun_out/r05/configs_l.log 2>&1; echo "exit $?"; grep -A30 "perf guard" gpurun_out/r05/configs_l.log | cut -c1-200
cp gpurun_out/configs.json gpurun_out/r05/configs_l.json
echo "### fuzz 240 s"
timeout 600 python tools/fuzz_parity.py 240 1111 2>&1 | tail -3
echo "### latency stamps, narrow shapes"
PIPE_SWEEPS=0 bash tools/blk_ab.sh stamps "4096 512 1024 1.585 3 0" "4096 1024 768 1.585 3 0" "784 128 512 4 5 0" 2>&1 | cut -c1-260
echo "### prof_round r05"
ROUND=r05 timeout 2400 bash tools/prof_round.sh > gpurun_out/r05/prof_round.log 2>&1; grep -A2 "gpfq_blk_kernel" gpurun_out/prof_r05/bench_kernel_stats.csv | head -3 | cut -c1-200
} >> $L 2>&1
tail -50 $L
