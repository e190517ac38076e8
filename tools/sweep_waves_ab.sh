echo "== headline: 11 vs 8 sweep wavefronts, symmetric (0) and general (2) forms"
PIPE_MODES=2 PIPE_VARIANTS=0,2 PIPE_SWEEPS=11,8 timeout 900 python tools/pipe_probe.py 4096 4096 1024 1.585 3 0 2>&1 | grep -E "pipe mode|Error" | cut -c1-200
echo "== m=512"
PIPE_MODES=2 PIPE_VARIANTS=0 PIPE_SWEEPS=11,8 timeout 900 python tools/pipe_probe.py 4096 4096 512 1.585 3 0 2>&1 | grep -E "pipe mode|Error" | cut -c1-200
