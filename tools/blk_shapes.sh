# usage (GPU box): bash tools/blk_shapes.sh -- the block-pipelined dense kernel (option pipe=2) against the kernels the dispatch would
# otherwise take (pipe=0), over the shapes that decide gpfq_capi.hip's `fits` rule
for sh in "4096 4096 1024" "4096 4096 512" "4096 4096 700" "4096 4096 1024 6 8" "4096 4096 2048 4 5" "4096 512 1024" "784 4096 512 4 5"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|rror"
done
