# usage (GPU box): bash tools/split_sweep.sh  -- pair splits from the environment (diagnostic build -DGPFQ_BLK_SPLIT_ENV):
# GPFQ_BLK_SPLIT for the 8-wavefront shapes (any counts up to one more than an even split), GPFQ_BLK_SPLIT11 for the 11-wavefront
# shapes (permutations of the shape's own counts)
export GPFQ_DIAG="-DGPFQ_BLK_SPLIT_ENV"
G='pipe mode|rror'
run() { # var shape splits...
  var="$1"; sh="$2"; shift; shift
  echo "== $sh"
  for sp in "$@"; do
    if [ "$sp" = "-" ]; then unset $var; else export $var="$sp"; fi
    echo -n "  split ${sp}: "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "$G" | sed -e 's/.*\]: //' | cut -c1-40
  done
}
run GPFQ_BLK_SPLIT11 "4096 4096 1024 1.585 3 0" - 23333333333 32333333333 33233333333 33323333333 33332333333 33333233333 33333323333 33333332333 33333333233 33333333323 33333333332 -
