#!/bin/bash
# GPU-box session: smoke, headline bench, rocprofv3 kernel stats of the same command.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench.err; cat gpurun_out/bench.json | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','model_build_s','greedy_ids_identical_across_steps')})
    print('roofline', {k: r['roofline'][k] for k in ('achieved','frac','avg_launch_us','launches')})
    print('roofline_prefill', {k: r['roofline_prefill'][k] for k in ('achieved','frac','avg_launch_us','prefill_end_to_end_frac')})
    print('cpu', r['cpu_baseline'])
    for k, v in r['kernel_breakdown_ms_per_step'].items(): print(f'  {k:26s} {v[\"ms\"]:10.3f} ms  n={v[\"n\"]}')
"
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/prof_bench.json 2> $OLDPWD/gpurun_out/prof.err); echo "rocprof rc=$?"
  mkdir -p gpurun_out/prof
  find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
  ls -la /tmp/prof | head -20
  head -25 gpurun_out/prof/*kernel_stats.csv 2>/dev/null
fi
