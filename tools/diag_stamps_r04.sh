# usage (GPU box): bash tools/diag_stamps_r04.sh -- in-kernel phase stamps of the headline shape, 8 and 11 sweep wavefronts (diagnostic build)
export GPFQ_DIAG="-DGPFQ_BLK_STAMPS"
for sh in "4096 4096 1024 1.585 3 0"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=${PIPE_SWEEPS:-11} timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|slot top|Error" | cut -c1-250
done
