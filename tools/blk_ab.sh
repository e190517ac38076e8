#!/bin/bash
# usage (GPU box, repository root): bash tools/blk_ab.sh <mode> [args]   -- A/B timings of the block-pipelined dense kernel (tools/pipe_probe.py), every
# variant bit-compared with the row-group kernel (and the oracle where a sample size is given).  One script instead of the eight of rounds 2-4:
#   shapes             the shapes behind gpfq_capi.hip's dispatch rule (round 2: blk_shapes.sh)
#   latency            the latency-bound shapes and the north-star layer's 2 / 4 / 8-GPU shards (round 3: latency_shapes.sh)
#   waves ["N C m bits scalar oracle" ...]   8 against 11 sweep wavefronts, symmetric and general forms (rounds 3-4: sweep_waves_ab.sh, mfma_ab.sh)
#   stamps ["shape" ...]                     the same with the in-kernel phase stamps (diagnostic build, rebuilt on the box; diag_stamps*.sh)
#   flags "<hipcc flags>" ...                kernel time of the headline shape under diagnostic builds ("" = shipped): -DGPFQ_BLK_NO_MFMA (dot products
#                                            on the vector unit), -DGPFQ_BLK_NO_FUSED (matrix-unit dot products as a phase of their own), -DGPFQ_BLK_STAMPS
#                                            (the in-kernel stamps): csrc/gpfq_blk_diag.hpp.  (The marginal-cost switches of rounds 4-5 -- a part of
#                                            the slot issued twice, the barrier removed -- were removed in round 6: profiles/r04, profiles/r05 hold their numbers.)
mode=$1; shift
G='old kernel|pipe mode|oracle|cycles per slot|decision wave|slot top|rror'
probe() { timeout 900 python tools/pipe_probe.py $1 2>&1 | grep -E "$G" | cut -c1-250; }
case $mode in
  shapes)
    for sh in "4096 256 1024" "4096 64 1024" "4096 10 1024" "4096 128 512" "4096 10 300" "2048 128 2048" "2048 16 1536" "300 10 1000"; do
      echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 probe "$sh"; done ;;
  latency)
    for sh in "4096 512 1024 1.585 3 16" "784 128 512 4 5 16" "2048 128 5008 3 4 8" "4096 4096 1024 1.585 3 16" "4096 2048 1024 1.585 3 16" "4096 1024 1024 1.585 3 16" \
              "4096 1000 2048 4 5 8" "4096 4096 2048 4 5 8" "4096 4096 768 1.585 3 8" "4096 4096 512 1.585 3 8" "4096 4096 5008 3 4 0"; do
      echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 probe "$sh"; done ;;
  waves)
    if [ $# -eq 0 ]; then set -- "4096 4096 1024 1.585 3 64" "4096 4096 768 1.585 3 16" "4096 4096 512 1.585 3 16" "4096 4096 1000 4 5 16"; fi
    for sh in "$@"; do echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=${PIPE_VARIANTS:-0,2} PIPE_SWEEPS=${PIPE_SWEEPS:-11,8} probe "$sh"; done ;;
  stamps)
    export GPFQ_DIAG="-DGPFQ_BLK_STAMPS"
    if [ $# -eq 0 ]; then set -- "4096 4096 1024 1.585 3 0"; fi
    for sh in "$@"; do echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=${PIPE_SWEEPS:-11,8} probe "$sh"; done ;;
  flags)
    for fl in "$@"; do
      export GPFQ_DIAG="$fl"; [ -z "$fl" ] && unset GPFQ_DIAG
      echo "== flags: $fl"; PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=${PIPE_SWEEPS:-0} probe "${SHAPE:-4096 4096 1024 1.585 3 0}"; done ;;
  *) echo "modes: shapes | latency | waves | stamps | flags"; exit 2 ;;
esac
