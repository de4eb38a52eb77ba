#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r2y; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_real_geometry_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do for p in 1 0; do
  LMX_ATTN_PREFETCH=$p timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-batch --no-pmc > $O/bench_pf${p}_$rep.json 2> $O/bench_pf${p}_$rep.err
  python - $O/bench_pf${p}_$rep.json $p <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r.get('kernel_breakdown_ms_per_step', {})
    print('prefetch=%s' % sys.argv[2], {k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s')}, 'ids_same', r.get('greedy_ids_identical_across_steps'),
          {k.replace('decode.',''): round(kb[k]['ms'] / max(1, kb[k]['n']) * 1e3, 1) for k in kb if k.startswith('decode.gemv') or k == 'decode.attn'})
PY
done; done
