#!/bin/bash
mkdir -p gpurun_out/r05
L=gpurun_out/r05/i.log
: > $L
{
echo "### decision wavefront: '' = quick cert + header prefetch behind the store wait, bare s_barrier; HDR_WAIT = round 4's barrier; last = round 4's decision wavefront"
for sh in "4096 512 1024 1.585 3 0" "4096 1024 1024 1.585 3 0" "4096 2048 1024 1.585 3 0" "4096 4096 1024 1.585 3 0" "784 128 512 4 5 0" "4096 1024 768 1.585 3 0" "4096 1000 2048 4 5 0" "2048 128 5008 3 4 0" "4096 4096 2048 4 5 0"; do
  echo "== shape $sh"
  for rep in 1 2; do
  for fl in "" "-DGPFQ_BLK_HDR_WAIT" "-DGPFQ_BLK_NO_QUICK_CERT -DGPFQ_BLK_HDR_WAIT -DGPFQ_BLK_TABLE_NEIGHBOURS"; do
    export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
    echo -n "  [$fl] "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror" | sed -e 's/.*\]: //' | cut -c1-110
  done; done
done
unset GPFQ_DIAG
echo "### parity"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fullsize_configs.py -m gpu -x -q -k "not conv" 2>&1 | tail -5
echo "### fuzz 120 s"
timeout 600 python tools/fuzz_parity.py 120 808 2>&1 | tail -3
} >> $L 2>&1
tail -80 $L
