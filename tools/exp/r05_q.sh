#!/bin/bash
# cluster form: the exchange's loads from the XCD's L2 first (default build) against device scope from the first poll (-DGPFQ_CL_FAR_ONLY)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/q2.log
: > $L
{
echo "### cluster form: parity (default build)"
timeout 900 python -m pytest tests/test_cluster_form_gpu.py -m gpu -x -q 2>&1 | tail -5
for fl in "" "-DGPFQ_CL_NEAR_INV"; do
  export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
  echo "### build [$fl]"
  for sh in "4096 4096 8192 1.585 3 64" "4096 4096 6000 4 5 32"; do
    echo "== shape $sh"
    PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-230
  done
  echo "== shape 4096 4096 5008 3 4 32, cluster threshold 4096"
  BLK_CLUSTER=4096 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 4096 5008 3 4 32 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-230
  echo "== shape 4096 4096 2048 4 5 32, cluster threshold 1024"
  BLK_CLUSTER=1024 PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py 4096 4096 2048 4 5 32 2>&1 | grep -E "pipe mode|rror|!!" | cut -c1-230
done
} >> $L 2>&1
tail -60 $L
