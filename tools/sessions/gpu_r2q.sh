#!/bin/bash
# Round-2 GPU session Q: persistent decode step: parity tests, then the headline bench with / without it.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2q; mkdir -p $O
export TMPDIR=/tmp
echo skip tests
for p in 1 0; do
  LMX_DECODE_PERSIST=$p timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-pmc > $O/bench_p$p.json 2> $O/bench_p$p.err
  python - $O/bench_p$p.json $p <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r.get('kernel_breakdown_ms_per_step', {})
    print('persist=%s' % sys.argv[2], {k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s') if k in r}, 'ids_same', r.get('greedy_ids_identical_across_steps'),
          {k: round(kb[k]['ms'] / max(1, kb[k]['n']) * 1e3, 1) for k in kb if k.startswith('decode.')})
PY
  tail -2 $O/bench_p$p.err | cut -c1-300
done
