#!/bin/bash
# Round 5, session I: the V^T cache as 64-key tiles ([s_max / 64][D][64] per head) — parity of every reader / writer, then the A/B numbers
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_qkv_fuse_gpu.py tests/test_vision_fuse_gpu.py tests/test_decode_splitq_gpu.py tests/test_batching_gpu.py tests/test_beam_gpu.py tests/test_reuse_gpu.py -q -x -p no:cacheprovider -n 4 2>&1 | tail -6 ) 2>&1
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection > gpurun_out/r05_bench_vt_tiles.json 2>> gpurun_out/r05_i.err
python tools/bench_brief.py gpurun_out/r05_bench_vt_tiles.json "V^T tiles" | head -4
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_vt_tiles.json')); print('serving_batch', {k: round(v['decode_tokens_per_s']) for k, v in d['serving_batch']['by_batch'].items()})"
for b in 8 32; do timeout 90 python tools/mb_tp_batch_step.py 1 $b 2>/dev/null | tee -a gpurun_out/r05_batch_step_vt_tiles.jsonl | cut -c1-400; done
timeout 100 python tools/mb_kv_attn.py 1150 2>/dev/null | tail -1
tail -3 gpurun_out/r05_i.err
