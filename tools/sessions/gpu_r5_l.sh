#!/bin/bash
# Round 5, session L: batch decode attention with two heads per workgroup (rotated wave assignment) — parity of the batch consumers, then the batched step
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_batching_gpu.py tests/test_beam_gpu.py tests/test_stop_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4 ) 2>&1
for b in 8 16 32; do timeout 90 python tools/mb_tp_batch_step.py 1 $b 2>/dev/null | tee -a gpurun_out/r05_batch_step_hpw2.jsonl | cut -c1-330; done
