# usage (GPU box): bash tools/blk_shapes.sh -- the block-pipelined dense kernel (option pipe=2) against the kernels the dispatch would
# otherwise take (pipe=0), over the shapes that decide gpfq_capi.hip's `fits` rule
for sh in "4096 256 1024" "4096 64 1024" "4096 10 1024" "4096 128 512" "4096 10 300" "2048 128 2048" "2048 16 1536" "300 10 1000"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|rror"
done
