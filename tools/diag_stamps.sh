export GPFQ_DIAG="-DGPFQ_BLK_STAMPS"
for sh in "4096 512 1024 1.585 3 0" "784 128 512 4 5 0" "2048 128 5008 3 4 0" "4096 1024 1024 1.585 3 0" "4096 4096 1024 1.585 3 0"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=8 timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|cycles per slot|decision wave|Error" | cut -c1-250
done
