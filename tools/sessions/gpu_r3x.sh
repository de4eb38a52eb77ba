#!/bin/bash
# Round-3 GPU session X: exactly what the driver runs at round end: smoke(), then the bench with its flags.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3x; mkdir -p $O
export TMPDIR=/tmp
SECONDS=0; timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; echo "smoke rc=$? (${SECONDS}s)"; tail -4 $O/smoke.txt
SECONDS=0; timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$? (${SECONDS}s)"
python - $O/bench_driver.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: (round(r[k], 3) if isinstance(r[k], float) else r[k]) for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','prefill_ms','decode_ms_per_token')})
    print('roofline', round(r['roofline']['frac'], 4), 'prefill', round(r['roofline_prefill']['frac'], 4), 'cpu', r['cpu_baseline']['value'], r['cpu_baseline']['kind'], r['cpu_baseline']['cores'])
PY
