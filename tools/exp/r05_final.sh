#!/bin/bash
# round 5, final validation of HEAD: the whole suite, bench, perf guard, fuzz, the end-to-end tools, the bench's kernel trace
mkdir -p gpurun_out/r05
L=gpurun_out/r05/final.log
: > $L
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
{
echo "### pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5
echo "### bench.py"
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-2500
echo "### bench_configs --resnet --shapes --check profiles/r05/configs.json"
timeout 2400 python tools/bench_configs.py --resnet --shapes --check profiles/r05/configs.json > gpurun_out/r05/configs_final.log 2>&1; echo "exit $?"; grep -A40 "perf guard" gpurun_out/r05/configs_final.log | cut -c1-200
cp gpurun_out/configs.json gpurun_out/r05/configs_final.json
echo "### fuzz 240 s"
timeout 900 python tools/fuzz_parity.py 240 3333 2>&1 | tail -3
echo "### e2e tools"
for t in "e2e_mlp.py" "e2e_cnn.py" "e2e_vgg16.py"; do echo "-- $t"; timeout 900 python tools/$t 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300; done
echo "### rocprofv3 --kernel-trace --stats -- python3 bench.py"
OUT=$ROOT/gpurun_out/prof_r05_final; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $ROOT/bench.py > $OUT/bench.log 2>&1)
find $OUT/bench -name "bench_kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
head -6 $OUT/bench_kernel_stats.csv | cut -c1-220
} >> $L 2>&1
tail -70 $L
