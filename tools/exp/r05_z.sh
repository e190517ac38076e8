#!/bin/bash
# pair splits of the four-group narrow shapes on rows of 769..1024 samples, re-measured after the sweep-side flush (diagnostic build -DGPFQ_BLK_SPLIT_ENV)
mkdir -p gpurun_out/r05
L=gpurun_out/r05/z.log
: > $L
{
export GPFQ_DIAG="-DGPFQ_BLK_SPLIT_ENV"
G='pipe mode|rror'
run() { var="$1"; sh="$2"; shift; shift
  echo "== $sh"
  for sp in "$@"; do
    if [ "$sp" = "-" ]; then unset $var; else export $var="$sp"; fi
    echo -n "  split ${sp}: "; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "$G" | sed -e 's/.*\]: //' | cut -c1-40
  done
}
run GPFQ_BLK_SPLIT "4096 512 1024 1.585 3 0" - 25542554 35443454 34453445 44444444 24552455 35543444 -
run GPFQ_BLK_SPLIT "4096 2048 1024 1.585 3 0" - 25542554 35443454 34453445 44444444 24552455 -
run GPFQ_BLK_SPLIT "4096 512 1000 4 5 0" - 35443454 44444444
echo "### cluster form, 8 neurons per workgroup: 2048 x 128 on 5008 samples"
BLK_CLUSTER=1 run GPFQ_BLK_SPLIT "2048 128 5008 3 4 0" - 35443454 44444444 34453445
} >> $L 2>&1
cat $L
