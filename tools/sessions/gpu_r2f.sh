#!/bin/bash
# Round-2 GPU session F: whole -m gpu suite, smoke, headline bench + rocprofv3 kernel stats of the same command.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2f; mkdir -p $O
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?"; tail -3 $O/smoke.log
echo "== gpu suite"; timeout 2400 python -m pytest tests -m gpu -q > $O/test_gpu.log 2>&1; echo "rc=$?"; tail -15 $O/test_gpu.log
echo "== bench"; timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 400 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2f/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline', {k: r['roofline'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','traffic','event_pair_overhead_us','event_pair_calibration')})
    print('roofline_prefill', {k: r['roofline_prefill'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','prefill_end_to_end_frac','by_shape_tflops')})
    print('cpu', {k: r['cpu_baseline'].get(k) for k in ('value','prefill_ms','decode_tokens_per_s','cores','cpu','measured')})
PY
echo "== rocprofv3 kernel stats of the bench command"
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o run -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof.err); echo "rocprof rc=$?"
mkdir -p $O/prof; find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/prof/ \; ; find /tmp/prof -name "*domain_stats.csv" -exec cp {} $O/prof/ \;
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r2f/prof/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), r["AverageNs"].rjust(12), r["Percentage"].rjust(7))
PY
