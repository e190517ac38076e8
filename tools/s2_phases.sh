#!/bin/bash
# usage (GPU box, repository root): bash tools/s2_phases.sh
# Timing experiments on gpfq_gram_s2_kernel: diagnostic builds that leave one phase out (sums are wrong: only the times count).
for k in 0 1 2 3 4; do
  export GPFQ_DIAG="-DGPFQ_S2_SKIP=$k"
  echo "== GPFQ_S2_SKIP=$k"
  python tools/conv1_probe.py 2>&1 | tail -1
  python tools/conv1_probe.py --first 2>&1 | tail -1
done
