#!/bin/bash
# Round-3 GPU session A: dataflow decode step — parity tests, then decode ms/token with / without it (tools/mb_decode.py), then the headline bench.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decode_flow_gpu.py -x -q > $O/pytest_flow.txt 2>&1; echo "pytest flow rc=$?"; tail -5 $O/pytest_flow.txt
timeout 600 python tools/mb_decode.py "" "LMX_DECODE_FLOW=0" "LMX_FLOW_TIMELINE=1" "LMX_FLOW_R_O=1,LMX_FLOW_R_DOWN=1" "LMX_FLOW_R_QKV=2,LMX_FLOW_R_GU=2" "LMX_FLOW_R_O=4,LMX_FLOW_R_DOWN=4" > $O/mb_decode.jsonl 2> $O/mb_decode.err; echo "mb rc=$?"; cat $O/mb_decode.jsonl; tail -3 $O/mb_decode.err
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > $O/pytest_more.txt 2>&1; echo "pytest more rc=$?"; tail -5 $O/pytest_more.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r.get('kernel_breakdown_ms_per_step', {})
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_tokens_per_s','decode_ms_per_token') if k in r}, 'ids_same', r.get('greedy_ids_identical_across_steps'))
    print('roofline', {k: r['roofline'][k] for k in ('achieved','frac','avg_launch_us')})
    print({k: round(kb[k]['ms'] / max(1, kb[k]['n']) * 1e3, 1) for k in kb if k.startswith('decode.')})
PY
tail -3 $O/bench.err | cut -c1-300
