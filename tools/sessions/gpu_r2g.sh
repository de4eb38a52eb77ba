#!/bin/bash
# Round-2 GPU session G: A/B of the padded-block skip in the ping-pong GEMM (same box, interleaved runs).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gemm8p_gpu.py -q -x > $O/test_gemm8p.log 2>&1; echo "gemm8p tests rc=$?"; tail -2 $O/test_gemm8p.log
for rep in 1 2; do
for ns in 0 1; do
  LMX_GEMM8P_NOSKIP=$ns timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-batch --no-pmc > $O/bench_ns${ns}_$rep.json 2> $O/bench_ns${ns}_$rep.err
  python - $O/bench_ns${ns}_$rep.json $ns <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r['kernel_breakdown_ms_per_step']
    print('noskip=%s' % sys.argv[2], {k: round(r[k], 3) for k in ('value','prefill_ms')}, 'frac', round(r['roofline_prefill']['frac'], 4),
          {k.split('.')[-1]: round(kb[k]['ms'] / kb[k]['n'] * 1e3, 1) for k in kb if k.startswith('prefill.gemm')}, 'ids_same', r['greedy_ids_identical_across_steps'])
PY
done; done
for ns in 0 1; do
echo "== microbench noskip=$ns"; LMX_GEMM8P_NOSKIP=$ns timeout 300 python tools/mb_gemm_variants.py "1087,12288,4096;1087,22016,4096;1087,4096,11008" "30,-1" 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['N'], r['K'], 'v', r['variant'], r.get('us_median'), r.get('TFs_median'))"
done
