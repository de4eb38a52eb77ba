#!/bin/bash
# Round 4, session u: hand-counted load stream in the decode batch's linear (skinny.hip STREAM): parity (stream == hipcc's waits, bit for bit), microbenchmark, serving batch A/B.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_batching_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 ) 2>&1
for sflag in 0 1; do LMX_SKINNY_STREAM=$sflag LMX_SKINNY_XNORM=1 timeout 300 python tools/mb_skinny_norm.py 2 8 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('STREAM=$sflag', r['shape'], 'M', r['M'], 'linear alone', r['us']['plain'], 'rmsnorm + linear', r['us']['unfused'])" | tee -a gpurun_out/r04_skinny_stream_mb.txt; done
for sflag in 0 1; do
  LMX_SKINNY_STREAM=$sflag timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection > gpurun_out/r04_u_bench_stream$sflag.json 2>> gpurun_out/r04_u.err
  python - "$sflag" <<'PY' | tee -a gpurun_out/r04_skinny_stream_serving_ab.jsonl
import json, sys
f = sys.argv[1]
for l in open(f"gpurun_out/r04_u_bench_stream{f}.json"):
    try: r = json.loads(l)
    except Exception: continue
    sb = r.get("serving_batch", {}).get("by_batch", {})
    print(json.dumps({"LMX_SKINNY_STREAM": int(f), "value": round(r["value"], 1), "serving_batch": {k: [round(v["ms_per_step"], 3), round(v["decode_tokens_per_s"])] for k, v in sb.items()}}))
PY
done
