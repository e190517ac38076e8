#!/bin/bash
# workgroup maps again, where the slice count divides 8: times (8- and 16-neuron workgroups) and FETCH_SIZE of the 8192-sample layer under either map
mkdir -p gpurun_out/r05
L=gpurun_out/r05/map.log
: > $L
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
{
for sh in "4096 1000 2048 4 5 16" "4096 2048 2048 4 5 16" "4096 512 4096 4 5 16" "4096 4096 4096 4 5 16" "4096 4096 8192 1.585 3 16" "4096 300 8192 1.585 3 16" "4096 4096 16384 1.585 3 8"; do
  echo "== shape $sh"
  for mp in 0 1; do
    echo -n "  map $mp "; BLK_CLUSTER_MAP=$mp PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=0 timeout 600 python tools/pipe_probe.py $sh 2>&1 | grep -E "pipe mode|rror|!!" | sed -e 's/.*sweeps=0 //' | cut -c1-150
  done
done
} >> $L 2>&1
cat $L
