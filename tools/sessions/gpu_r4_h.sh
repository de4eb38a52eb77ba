#!/bin/bash
# Round 4, session H: one-workgroup-per-head decode attention (parity + in-situ A/B).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "decode_attn_head or decode_attn_flow" -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r04_h_ops.log; tail -5 gpurun_out/r04_h_ops.log
LMX_ATTN_HEAD=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_decode_flow_gpu.py tests/test_full_depth_gpu.py -m gpu -q -x -k "not fp32_engine_full" -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/r04_h_model.log; tail -4 gpurun_out/r04_h_model.log
for v in 1 0 1 0; do
  LMX_ATTN_HEAD=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_h_bench_head$v.json 2> gpurun_out/r04_h_bench.err || tail -5 gpurun_out/r04_h_bench.err
  python tools/bench_brief.py gpurun_out/r04_h_bench_head$v.json "ATTN_HEAD=$v" | sed -n '1p;3p'
done
