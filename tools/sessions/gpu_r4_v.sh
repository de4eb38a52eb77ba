#!/bin/bash
# Round 4, session v2: the hand-counted stream of skinny.hip without past-the-end loads (exact tail counts).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_batching_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2 ) 2>&1
LMX_SKINNY_STREAM=1 LMX_SKINNY_XNORM=1 timeout 300 python tools/mb_skinny_norm.py 2 8 16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('STREAM=1c', r['shape'], 'M', r['M'], 'linear alone', r['us']['plain'], 'rmsnorm + linear', r['us']['unfused'])" | tee -a gpurun_out/r04_skinny_stream_mb.txt
LMX_SKINNY_STREAM=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection > gpurun_out/r04_v_bench_stream1.json 2>> gpurun_out/r04_v.err
python - <<'PY' | tee -a gpurun_out/r04_skinny_stream_serving_ab.jsonl
import json
for l in open("gpurun_out/r04_v_bench_stream1.json"):
    try: r = json.loads(l)
    except Exception: continue
    sb = r.get("serving_batch", {}).get("by_batch", {})
    print(json.dumps({"LMX_SKINNY_STREAM": "1 (no dummies)", "value": round(r["value"], 1), "serving_batch": {k: [round(v["ms_per_step"], 3), round(v["decode_tokens_per_s"])] for k, v in sb.items()}}))
PY
