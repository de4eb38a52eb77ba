#!/bin/bash
# Round 4, session D: turn-to-turn reuse — parity tests, then the config-4 harness with and without it.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reuse_gpu.py tests/test_stop_gpu.py tests/test_batching_gpu.py tests/test_tool_loop_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r04_d_tests.log; tail -12 gpurun_out/r04_d_tests.log
for r in 0 1; do
  timeout 900 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 --reuse $r > gpurun_out/r04_d_config4_reuse$r.json 2> gpurun_out/r04_d_config4.err || tail -20 gpurun_out/r04_d_config4.err
  python - <<P
import json
for l in open("gpurun_out/r04_d_config4_reuse$r.json"):
    try: d = json.loads(l)
    except Exception: continue
    print("reuse=$r", {k: d.get(k) for k in ("completed", "wall_s", "generated_tokens_per_s", "median_ttft_s", "median_round2_ttft_s", "median_total_s", "answers_as_scripted", "reuse", "errors")})
P
done
