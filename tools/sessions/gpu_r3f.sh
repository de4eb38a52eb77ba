#!/bin/bash
# Round-3 GPU session F: TP serving / batching / train tests after the symmetric-failure protocol, then the FULL default bench (CPU baseline incl. fp32, tp_projection, serving batch).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
grep -E "passed|failed" $O/pytest.txt 2>/dev/null | tail -1
SECONDS=0; timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; echo "bench wall ${SECONDS}s"
python - $O/bench.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s','decode_ms_per_token') if k in r})
    print('cpu', {k: r['cpu_baseline'].get(k) for k in ('value','cores','prefill_ms','decode_tokens_per_s','decode_steps_timed','partly_priced','fp32')} if r.get('cpu_baseline') else None)
    tp = r.get('tp_projection') or {}
    print('tp', {w: {k: round(v[k], 3) for k in ('rank_compute_prefill_ms','rank_compute_decode_ms_per_token','projected_prefill_ms','projected_decode_ms_per_token','projected_value_tokens_per_s','projected_speedup_vs_tp1')} for w, v in tp.get('by_world', {}).items()} if 'by_world' in tp else tp)
    print('serving', r.get('serving_batch', {}).get('by_batch'))
    print('traffic', r['roofline'].get('traffic'), (r['roofline'].get('traffic_source') or {}).get('ratio'))
PY
