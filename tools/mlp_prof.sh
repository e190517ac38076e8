#!/bin/bash
# usage (GPU box): tools/mlp_prof.sh   -- kernel trace of tools/e2e_mlp.py (the reference's MNIST MLP run) under gpurun_out/mlp_prof
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mlp_prof -o mlp -- python3 $GRAFT_REPO_ROOT/tools/e2e_mlp.py 25000 500,300 1.585 > $GRAFT_REPO_ROOT/gpurun_out/mlp_prof.log 2>&1
