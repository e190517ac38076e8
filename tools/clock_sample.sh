#!/bin/bash
# rocm-smi's shader clock and package power sampled while bench.py runs 1500 steps (diagnostic, round 6): gpurun -- 'bash tools/clock_sample.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --gpus 1 --steps 1500 --warmup 5 --numpy-sample 0 --long-rows 0 --cpu-sample 0 > /tmp/b.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done | sort | uniq -c | sort -k1,1nr | head -12
python -c "
import json; o=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0]); print('ms_per_step', o['ms_per_step'], 'kernel', o['roofline']['kernel_ms_avg'])"
