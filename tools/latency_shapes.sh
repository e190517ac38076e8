# usage (GPU box): bash tools/latency_shapes.sh -- the latency-bound dense shapes VERDICT r02 names (kernel times, bit-compared with the
# row-group kernel): 4096 x 512 on 1024 samples ternary, cfg1 Dense(784 -> 128) 512 samples 16 levels, cfg4 Dense(2048 -> 128) 5008 samples 8 levels,
# the headline shape and its 2 / 4 / 8-GPU shards
for sh in "4096 512 1024 1.585 3 16" "784 128 512 4 5 16" "2048 128 5008 3 4 8" "4096 4096 1024 1.585 3 16" "4096 2048 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" "4096 4096 2048 4 5 8" "4096 4096 512 1.585 3 8"; do
  echo "== $sh"; PIPE_MODES=${PIPE_MODES:-2} PIPE_VARIANTS=0 PIPE_SWEEPS=8 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|oracle|cycles per slot|decision wave|Error" | cut -c1-250
done
echo "== same shapes without the two-neuron workgroups"
for sh in "4096 512 1024 1.585 3 0" "784 128 512 4 5 0" "2048 128 5008 3 4 0"; do
  echo "== $sh (blk_pair_groups=0)"; BLK_PAIRS=0 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=8 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|Error" | cut -c1-250
done
