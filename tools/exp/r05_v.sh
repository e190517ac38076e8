#!/bin/bash
# validation with the cluster form dispatched by default: the whole suite, bench, perf guard, fuzz
mkdir -p gpurun_out/r05
L=gpurun_out/r05/v.log
: > $L
{
echo "### pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
echo "### bench.py"
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1200
echo "### bench_configs --resnet --shapes --check profiles/r05/configs.json"
timeout 2400 python tools/bench_configs.py --resnet --shapes --check profiles/r05/configs.json > gpurun_out/r05/configs_v.log 2>&1; echo "exit $?"; grep -A40 "perf guard" gpurun_out/r05/configs_v.log | cut -c1-200
cp gpurun_out/configs.json gpurun_out/r05/configs_v.json
echo "### fuzz 300 s"
timeout 900 python tools/fuzz_parity.py 300 2222 2>&1 | tail -3
} >> $L 2>&1
tail -60 $L
