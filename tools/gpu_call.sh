#!/bin/bash
# One parameterised GPU call (round 6; rounds 4-5 kept one script per call under tools/exp/):
#     gpurun --timeout 2400 -- 'bash tools/gpu_call.sh <step> [<step> ...]'
# runs the named steps in order on the GPU box, from the repository root, everything it writes under gpurun_out/r06/ (merged back by gpurun).
#   suite [pytest -k expr]   the whole -m gpu test suite (or a -k selection), tail of the report -> gpu_suite.txt
#   bench [args]             python bench.py [args] -> bench.json (+ the line's headline figures printed)
#   ab                       bench.py of this tree against the same command in _base_r05/ (a git worktree of the round's base), twice each
#   guard                    tools/bench_configs.py --skip-fc1 --shapes --check profiles/r05/configs.json (the perf guard) -> configs_check.txt
#   configs                  tools/bench_configs.py --resnet --shapes -> gpurun_out/configs.json
#   fuzz [seconds]           tools/fuzz_parity.py -> fuzz.txt
#   trace <tag> <script...>  rocprofv3 --kernel-trace --stats (csv; never the database format, whose post-processing hung a box for 39 minutes)
#                            of `python3 <script...>`, first rows of the per-kernel summary printed -> trace_<tag>_kernel_stats.csv
#   pmc <kernel> <script...> tools/pmc.sh: the counter passes over any probe
#   step                     tools/step_probe.py: a Dense layer's passes outside the recurrence kernel
# Steps with arguments end at the next step name or at `--`.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/r06
mkdir -p $OUT
show() { python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        o = json.loads([l for l in open(f) if l.startswith('{')][0])
        r = o['roofline']
        print(f"{f}: ms_per_step {o['ms_per_step']:.4f}  prefetched {o.get('ms_per_step_medians_prefetched')}  kernel {r['kernel_ms_avg']:.4f}  call {r['call_ms_avg']:.4f}"
              f"  frac {r['frac']:.4f}  value {o['value']:.4g}  parity {o.get('parity_sample')}  status {o.get('deferred_status_nonzero_steps')}")
    except Exception as e:
        print(f, 'ERR', e)
PY
}
is_step() { case "$1" in suite|bench|ab|guard|configs|fuzz|trace|pmc|step|--) return 0;; *) return 1;; esac; }
while [ $# -gt 0 ]; do
  step=$1; shift
  args=()
  # (the tag of `trace` / the kernel substring of `pmc` is taken as it comes, whatever it is called)
  if [ "$step" == "trace" ] || [ "$step" == "pmc" ]; then args+=("$1"); shift; fi
  while [ $# -gt 0 ] && ! is_step "$1"; do args+=("$1"); shift; done
  [ "$1" == "--" ] && shift
  echo "=== $step ${args[*]}"
  case $step in
    suite)
      if [ ${#args[@]} -gt 0 ]; then timeout 3000 python -m pytest tests -x -q -m gpu -k "${args[*]}" 2>&1 | tail -12 | tee $OUT/gpu_suite.txt
      else timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $OUT/gpu_suite.txt; fi ;;
    bench)
      timeout 900 python bench.py "${args[@]}" > $OUT/bench.json.tmp 2> $OUT/bench.err && mv $OUT/bench.json.tmp $OUT/bench.json; tail -2 $OUT/bench.err; show $OUT/bench.json ;;
    ab)
      for rep in 1 2; do
        (cd _base_r05 && timeout 600 python bench.py --steps 20 --warmup 3 --numpy-sample 0 --long-rows 0) > $OUT/bench_base_$rep.json 2> $OUT/bench_base_$rep.err
        timeout 600 python bench.py --steps 20 --warmup 3 --numpy-sample 0 --long-rows 0 > $OUT/bench_new_$rep.json 2> $OUT/bench_new_$rep.err
      done
      show $OUT/bench_base_1.json $OUT/bench_new_1.json $OUT/bench_base_2.json $OUT/bench_new_2.json ;;
    guard)
      timeout 1200 python tools/bench_configs.py --skip-fc1 --shapes --check profiles/r05/configs.json > $OUT/configs_check.txt 2>&1; grep -A40 "perf guard" $OUT/configs_check.txt ;;
    configs)
      timeout 2400 python tools/bench_configs.py --resnet --shapes > $OUT/configs.txt 2>&1; tail -5 $OUT/configs.txt ;;
    fuzz)
      timeout 1500 python tools/fuzz_parity.py ${args[0]:-300} 2>&1 | tail -6 | tee $OUT/fuzz.txt ;;
    trace)
      tag=${args[0]}
      if [ ${#args[@]} -lt 2 ]; then echo "trace: <tag> <script> [args]"; continue; fi
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o t -- python3 "${args[@]:1}" > $OUT/trace_$tag.log 2>&1
      find $OUT/trace_$tag -name "t_kernel_stats.csv" -exec cp {} $OUT/trace_${tag}_kernel_stats.csv \;
      find $OUT/trace_$tag -name "t_kernel_trace.csv" -exec cp {} $OUT/trace_${tag}_kernel_trace.csv \;
      tail -2 $OUT/trace_$tag.log | cut -c1-300; head -14 $OUT/trace_${tag}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-170 ;;
    pmc)
      timeout 2400 bash tools/pmc.sh "${args[@]}" 2>&1 | tee $OUT/pmc_${args[0]}.txt | tail -80 ;;
    step)
      timeout 300 python tools/step_probe.py "${args[@]}" 2>&1 | tee $OUT/step_probe.txt ;;
  esac
done
