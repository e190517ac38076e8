# usage (GPU box): bash tools/diag_x.sh "<flags>" ... -- timing experiments on the headline shape: diagnostic builds that double a part of the slot
# (same results: a part of the slot is issued TWICE and the difference is its marginal cost)
for fl in "$@"; do
  export GPFQ_DIAG="-DGPFQ_BLK_STAMPS $fl"
  echo "== flags: $fl"
  PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=${PIPE_SWEEPS:-11} timeout 900 python tools/pipe_probe.py 4096 4096 1024 1.585 3 0 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|slot top|Error" | cut -c1-250
done
