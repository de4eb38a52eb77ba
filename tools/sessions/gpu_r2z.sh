#!/bin/bash
# Round-2 GPU session Z: scheduler-side packed prefill: serving tests + config 4 harness (TP=1, TP=2 shared GPU) + serve_bench.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2z; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_batching_gpu.py tests/test_tp_serving_gpu.py tests/test_worker_flow_gpu.py tests/test_tool_loop_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
for pp in 1 0; do
  LLAVA_MI355X_PACKED_PREFILL=$pp timeout 600 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 --packed $pp > $O/config4_pp$pp.log 2> $O/config4_pp$pp.err
  grep '^{' $O/config4_pp$pp.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('packed=$pp', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('completed', 'wall_s', 'generated_tokens_per_s', 'median_ttft_s', 'median_total_s', 'answers_as_scripted')})"
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/config4_harness.py \
    --model llava_plus_v0_13b --requests 32 --batch 32 --shared-gpu > $O/config4_tp2.log 2> $O/config4_tp2.err
grep '^{' $O/config4_tp2.log | cut -c1-400
