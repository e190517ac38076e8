# usage (GPU box): bash tools/mfma_ab.sh [shapes...] -- round 4: the block kernel's dot products on the matrix unit (v_mfma_f64_4x4x4_4b_f64), 8 and 11 sweep
# wavefronts, symmetric (variant 0) and general (variant 2) forms, bit-compared with the row-group kernel and the oracle
if [ $# -eq 0 ]; then set -- "4096 4096 1024 1.585 3 64" "4096 4096 768 1.585 3 16" "4096 4096 512 1.585 3 16" "4096 4096 1000 4 5 16"; fi
for sh in "$@"; do
  echo "== $sh"; PIPE_MODES=2 PIPE_VARIANTS=${PIPE_VARIANTS:-0,2} PIPE_SWEEPS=${PIPE_SWEEPS:-11,8} timeout 900 python tools/pipe_probe.py $sh 2>&1 | grep -E "old kernel|pipe mode|oracle|cycles per slot|decision wave|Error" | cut -c1-250
done
