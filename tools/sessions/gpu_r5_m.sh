#!/bin/bash
# Round 5, session M: the packed prefills on their own thread / stream: scheduler tests, then config 4 with the prefill thread on / off
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_batching_gpu.py tests/test_stop_gpu.py tests/test_sampling_gpu.py tests/test_tool_loop_gpu.py tests/test_worker_flow_gpu.py tests/test_reuse_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ) 2>&1 | grep -E "passed|failed|error|real|Error" 
for pt in 2 1 0 2 1 0; do
  timeout 400 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 --reuse 1 --prefill-thread $pt > gpurun_out/r05_m_config4_pt$pt.json 2> gpurun_out/r05_m.err || tail -5 gpurun_out/r05_m.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r05_m_config4_pt$pt.json'))
print('prefill_thread=$pt', {k: d[k] for k in ('generated_tokens_per_s', 'wall_s', 'answers_as_scripted', 'ttft_ms', 'round2_ttft_ms', 'round1_ms', 'round2_ms', 'scheduler')})
PY
  cat gpurun_out/r05_m_config4_pt$pt.json >> gpurun_out/r05_m_config4_ab.jsonl
done
