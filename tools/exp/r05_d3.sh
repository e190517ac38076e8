#!/bin/bash
# cluster form, 4 / 8 neurons per workgroup: D three slots old (the exchange gathered a slot after its publish)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/d3.log
: > $L
{
echo "### parity"
timeout 1200 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "boundaries or role_split or two_neuron or one_neuron" 2>&1 | tail -3
echo "### timings"
for sh in "2048 128 5008 3 4 16" "128 10 5008 3 4 8" "4096 512 4096 4 5 16" "4096 1000 2048 4 5 16" "4096 2048 2048 4 5 16" "4096 1000 3000 4 5 16" "4096 300 8192 1.585 3 16" "4096 128 3000 4 5 16"; do
  echo "== shape $sh"
  for nl in 0 4 2 1; do
    echo -n "  BLK_CLUSTER_NL=$nl "; BLK_CLUSTER_NL=$nl PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
cat $L
