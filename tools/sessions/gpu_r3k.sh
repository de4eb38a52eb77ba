#!/bin/bash
# Round-3 GPU session K: evidence for profiles/ on the final kernels: full GPU suite, default bench, rocprofv3 kernel stats of the bench command, PMC passes.
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? (${SECONDS}s)"; tail -3 $O/pytest_gpu.txt
SECONDS=0; timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - $O/bench.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: round(r[k], 3) for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token') if k in r})
    print('roofline', {k: r['roofline'].get(k) for k in ('achieved','frac','traffic')}, 'prefill', {k: r['roofline_prefill'].get(k) for k in ('achieved','frac','by_shape_tflops')})
    print('cpu', {k: r['cpu_baseline'].get(k) for k in ('value','cores','decode_steps_timed','partly_priced')} if r.get('cpu_baseline') else None)
PY
cd /tmp; rm -rf /tmp/prof_r3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r3 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err
echo "rocprof bench rc=$?"
f=$(find /tmp/prof_r3 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python - "$f" $O/r03_rocprofv3_kernel_stats.csv <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo)
    for r in rows:
        r[0] = r[0][:160]
        w.writerow(r)
print("".join(",".join(r)[:200] + "\n" for r in rows[:16]))
PY
else echo "no kernel_stats.csv"; tail -5 $O/bench_under_rocprofv3.err; fi
cd $R
bash tools/gpu_pmc_r3.sh 2>&1 | tail -70
cp gpurun_out/r3pmc/r03_pmc*.json gpurun_out/r3pmc/r03_pmc_*.csv $O/ 2>/dev/null
timeout 300 python tools/mb_decode.py "" --tokens 64 2>&1 | tail -1 > $O/mb_decode.txt; cat $O/mb_decode.txt
