# usage (GPU box): bash tools/blk_shapes.sh -- the block-pipelined dense kernel (option pipe=2) against the kernels the dispatch would
# otherwise take (pipe=0), over the shapes that decide gpfq_capi.hip's `fits` rule
for sh in "784 4096 512 4 5" "4096 4096 512" "4096 4096 300" "4096 512 400" "4096 4096 700" "4096 4096 768 4 5" "4096 4096 1300" "4096 1024 1536 4 5" "1000 600 260"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|rror"
done
