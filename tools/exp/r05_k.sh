#!/bin/bash
# round 5, GPU call K: final decision wavefront (quick certification, split store wait, seven sweep wavefronts on rows <= 768 samples):
# two-neurons-per-lane shapes with 7 / 8 wavefronts, the whole suite, bench, configs + shapes against the perf guard, fuzz
mkdir -p gpurun_out/r05
L=gpurun_out/r05/k.log
: > $L
{
echo "### blk_quad_waves 7 / 8 on the two-neurons-per-lane narrow shapes (and one-neuron ones again)"
for sh in "4096 2048 768 1.585 3 0" "4096 2048 512 1.585 3 0" "4096 1500 600 4 5 0" "4096 1024 768 1.585 3 0" "4096 300 400 1.585 3 0"; do
  echo "== shape $sh"
  for rep in 1 2; do for nw in 7 8; do
    echo -n "  [blk_quad_waves $nw] "; BLK_QUAD_NW=$nw PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done; done
done
echo "### pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "### bench.py"
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1800
echo "### bench_configs --resnet --shapes --check profiles/r05/configs.json"
timeout 1800 python tools/bench_configs.py --resnet --shapes --check profiles/r05/configs.json > gpurun_out/r05/configs_k.log 2>&1; echo "exit $?"; grep -A30 "perf guard" gpurun_out/r05/configs_k.log | cut -c1-200
cp gpurun_out/configs.json gpurun_out/r05/configs_k.json
echo "### fuzz 200 s"
timeout 600 python tools/fuzz_parity.py 200 1010 2>&1 | tail -3
} >> $L 2>&1
tail -60 $L
