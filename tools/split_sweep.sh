# usage (GPU box): bash tools/split_sweep.sh  -- pair splits of the 8-wavefront shapes from the environment (diagnostic build -DGPFQ_BLK_SPLIT_ENV)
export GPFQ_DIAG="-DGPFQ_BLK_SPLIT_ENV"
G='pipe mode|rror'
run() { # shape splits...
  sh="$1"; shift
  echo "== $sh"
  for sp in "$@"; do
    if [ "$sp" = "-" ]; then unset GPFQ_BLK_SPLIT; else export GPFQ_BLK_SPLIT="$sp"; fi
    echo -n "  split ${sp}: "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "$G" | sed -e 's/.*\]: //' | cut -c1-40
  done
}
run "4096 1000 2048 4 5 8" - 12232222 12231223 13221322 12321232 12232222 11331133
run "4096 1024 1536 4 5 8" - 11221122 12211221 21122112
run "4096 2048 2048 4 5 8" - 25542554 15551555
