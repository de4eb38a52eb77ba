#!/bin/bash
# Round 4, session q: RMSNorm of the input rows inside the decode batch's linear (skinny.hip XN): parity, microbenchmark of both register forms, serving batch A/B;
# two-shot all-reduce at the real width between two processes (sizes that two ranks can keep resident on one GPU).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_batching_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5 ) 2>&1
( time timeout 400 python -m pytest tests/test_tp_p2p_gpu.py -q -x -s -k real_width -p no:cacheprovider 2>&1 | tail -6 ) 2>&1
for f in 1 2; do LMX_SKINNY_XNORM=$f timeout 300 python tools/mb_skinny_norm.py 2 8 16 >> gpurun_out/r04_skinny_xnorm_mb.jsonl 2>> gpurun_out/r04_q.err; done
cat gpurun_out/r04_skinny_xnorm_mb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shape'], 'M', r['M'], 'form', r['xnorm_form'], r['us'], 'saves', r['fused_saves_us'])"
for f in 0 1 2; do
  LMX_SKINNY_XNORM=$f timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection > gpurun_out/r04_q_bench_xn$f.json 2>> gpurun_out/r04_q.err
  python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
for l in open(f"gpurun_out/r04_q_bench_xn{f}.json"):
    try: r = json.loads(l)
    except Exception: continue
    sb = r.get("serving_batch", {}).get("by_batch", {})
    print("XNORM", f, "value", round(r["value"], 1), {k: (round(v["ms_per_step"], 3), round(v["decode_tokens_per_s"])) for k, v in sb.items()})
PY
done
tail -5 gpurun_out/r04_q.err
