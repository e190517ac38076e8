cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_device_alphabet_gpu.py -x -q 2>&1 | tail -3
timeout 900 python tools/bench_configs.py --skip-fc1 --shapes --check profiles/r05/configs.json > gpurun_out/r06/configs_check.txt 2>&1; grep -A30 "perf guard" gpurun_out/r06/configs_check.txt
timeout 600 python bench.py --steps 20 --warmup 3 --numpy-sample 0 --long-rows 0 > gpurun_out/r06/bench_new_1.json 2>gpurun_out/r06/bench_new_1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/bench_new_1.json')):
    o=json.loads([l for l in open(f) if l.startswith('{')][0])
    print(f, 'ms_per_step %.4f'%o['ms_per_step'], 'prefetched %s'%o.get('ms_per_step_medians_prefetched'), 'kernel %.4f'%o['roofline']['kernel_ms_avg'], 'call %.4f'%o['roofline']['call_ms_avg'], 'value %.4g'%o['value'], o.get('parity_sample'), o.get('deferred_status_nonzero_steps'))
PY
