#!/bin/bash
# Round-3 GPU session E: the headline bench (short) + per-kernel breakdown.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/${OUT:-r3e}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-pmc --no-batch > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r.get('kernel_breakdown_ms_per_step', {})
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s','decode_ms_per_token') if k in r}, 'ids_same', r.get('greedy_ids_identical_across_steps'))
    print('roofline', {k: (round(r['roofline'][k], 4) if isinstance(r['roofline'][k], float) else r['roofline'][k]) for k in ('achieved','frac','avg_launch_us')})
    rp = r['roofline_prefill']
    print('roofline_prefill', round(rp['frac'], 4), 'e2e', round(rp['prefill_end_to_end_frac'], 4), {k.split('.')[-1]: round(v, 1) for k, v in rp['by_shape_tflops'].items()})
    print('prefill ms:', {k: round(kb[k]['ms'], 3) for k in kb if not k.startswith('decode.')})
    print('decode us/launch:', {k: round(kb[k]['ms'] / max(1, kb[k]['n']) * 1e3, 2) for k in kb if k.startswith('decode.')})
PY
tail -3 $O/bench.err | cut -c1-300
