# usage (GPU box): bash tools/split_sweep.sh  -- pair splits from the environment (diagnostic build -DGPFQ_BLK_SPLIT_ENV):
# GPFQ_BLK_SPLIT for the 8-wavefront shapes (any counts up to one more than an even split), GPFQ_BLK_SPLIT11 for the 11-wavefront
# shapes (counts from two below the shape's largest up to it).  Edit the run lines for the shape at hand; the sweeps of round 4 are
# recorded in profiles/r04/README.md and in gpfq_blk.hip next to the tables they produced.
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
run GPFQ_BLK_SPLIT "4096 4096 768 1.585 3 0" - 24332433 14431443 23342334 13441344 24422442 -
run GPFQ_BLK_SPLIT "4096 4096 512 1.585 3 0" - 13221322 12321232 22222222 12231223 -
run GPFQ_BLK_SPLIT "4096 2048 1536 4 5 0" - 24332433 14431443 13441344
run GPFQ_BLK_SPLIT "4096 4096 1536 4 5 0" -
