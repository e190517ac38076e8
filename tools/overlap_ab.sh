#!/bin/bash
# bench.py with and without the second stream, alternating on one box (round 6): gpurun -- 'bash tools/overlap_ab.sh [reps]'
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06
for rep in $(seq 1 ${1:-3}); do
for v in "--no-overlap" "--overlap"; do
  python bench.py --steps 20 --warmup 3 --numpy-sample 0 --long-rows 0 $v 2>/dev/null | python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=o['roofline']
print('%-13s ms_per_step %.4f  medians prefetched %.4f  kernel %.4f  step - kernel %.3f  mismatches %s  status %s'%('$v',o['ms_per_step'],o.get('ms_per_step_medians_prefetched') or 0,r['kernel_ms_avg'],o['ms_per_step']-r['kernel_ms_avg'],(o.get('parity_sample') or {}).get('neurons_with_index_mismatch'),o.get('deferred_status_nonzero_steps')))"
done; done 2>&1 | tee gpurun_out/r06/overlap_ab.txt
