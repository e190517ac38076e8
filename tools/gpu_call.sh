cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/test_fullsize_cfg4_cfg5.py tests/test_fullsize_configs.py -x -q -k "whole_tensor or fc1" --durations=12 2>&1 | tail -25
