#!/bin/bash
# First GPU session of round 5: the streaming decode attention of csrc/attention_batch.h (written at the end of round 4, never run) —
# parity, then the batched step and the single request's decode step with and without it.  Nothing here is on by default.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 1 2; do
  echo "== LMX_BATCH_ATTN=$m: parity of the decode batch"
  ( LMX_BATCH_ATTN=$m timeout 240 python -m pytest tests/test_batching_gpu.py -q -x -p no:cacheprovider -k "batched or continuous or golden" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 ) 2>&1
done
for b in 8 32; do for m in 0 1 2; do
  LMX_BATCH_ATTN=$m timeout 90 python tools/mb_tp_batch_step.py 1 $b 2>/dev/null | sed "s/^{/{\"LMX_BATCH_ATTN\": $m, /" | tee -a gpurun_out/r05_batch_attn_wave.jsonl
done; done
for m in 1 2; do
  echo "== LMX_ATTN_WAVE=$m: parity of the single request (goldens)"
  ( LMX_ATTN_WAVE=$m timeout 300 python -m pytest tests/test_model_gpu.py -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 ) 2>&1
done
for m in 0 1 2; do
  echo "== LMX_ATTN_WAVE=$m"
  LMX_ATTN_WAVE=$m timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > gpurun_out/r05_bench_attn_wave$m.json 2>> gpurun_out/r05_a.err
  python tools/bench_brief.py gpurun_out/r05_bench_attn_wave$m.json "LMX_ATTN_WAVE=$m" | head -4
done
