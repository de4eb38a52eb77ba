#!/bin/bash
# Round 5, session Q: the decode loop stands back while a burst's backlog is prefilled (r5-H addendum): scheduler tests, the 32-request burst, config 4
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_batching_gpu.py tests/test_stop_gpu.py tests/test_sampling_gpu.py tests/test_tool_loop_gpu.py tests/test_worker_flow_gpu.py tests/test_reuse_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 ) 2>&1 | grep -E "passed|failed|rror"
SKIP_PLAIN=1 ONLY=packed_prefill,packed_prefill_between_steps timeout 600 python tools/serve_bench.py 32 32 128 > gpurun_out/r05_q_serve_bench.json 2> gpurun_out/r05_q.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_q_serve_bench.json'))
for k, v in d.items():
    if isinstance(v, dict): print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
for r in 1 1; do
timeout 400 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 --reuse 1 > gpurun_out/r05_q_config4.json 2>> gpurun_out/r05_q.err || tail -5 gpurun_out/r05_q.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_q_config4.json'))
print({k: d[k] for k in ('generated_tokens_per_s', 'answers_as_scripted', 'ttft_ms', 'round2_ttft_ms', 'scheduler')})
PY
cat gpurun_out/r05_q_config4.json >> gpurun_out/r05_q_config4.jsonl
done
