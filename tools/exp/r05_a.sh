#!/bin/bash
# round 5, GPU call A: baseline of this box, the free-running upper bound (no slot barrier: WRONG results, time only), stamps,
# eleven-wavefront pair splits, the new whole-layer oracle tests and the bench line with the NumPy pool baseline.
mkdir -p gpurun_out/r05
L=gpurun_out/r05/a.log
: > $L
{
echo "### flags: shipped vs NOBAR"
bash tools/blk_ab.sh flags "" "-DGPFQ_BLK_X_NOBAR" ""
echo "### stamps"
bash tools/blk_ab.sh flags "-DGPFQ_BLK_STAMPS"
echo "### splits (eleven sweep wavefronts, 4096x4096x1024)"
export GPFQ_DIAG="-DGPFQ_BLK_SPLIT_ENV"
for sp in - 33333333233 33333333323 33333333332 33323333333 33333332333; do
  if [ "$sp" = "-" ]; then unset GPFQ_BLK_SPLIT11; else export GPFQ_BLK_SPLIT11=$sp; fi
  echo -n "  split11 $sp: "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 4096 1024 1.585 3 0 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-120
done
unset GPFQ_DIAG GPFQ_BLK_SPLIT11
echo "### whole-layer tests"
timeout 1500 python -m pytest tests/test_fullsize_configs.py -m gpu -x -q -s -k "cfg2 or cfg3" 2>&1 | tail -15
echo "### bench"
timeout 900 python bench.py 2>&1 | tail -3
} >> $L 2>&1
tail -60 $L
