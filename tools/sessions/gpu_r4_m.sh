#!/bin/bash
# Round 4, session M: q|k|v RoPE epilogue with the table loads batched ahead of the stores (parity + bench).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qkv_fuse_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r04_m_tests.log; tail -3 gpurun_out/r04_m_tests.log
for i in 1 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_m_bench.json 2> gpurun_out/r04_m_bench.err || tail -5 gpurun_out/r04_m_bench.err
  python tools/bench_brief.py gpurun_out/r04_m_bench.json "batched rope table loads" | head -2 | cut -c1-420
done
LMX_FUSE_ROPE=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_m_bench0.json 2> gpurun_out/r04_m_bench.err
python tools/bench_brief.py gpurun_out/r04_m_bench0.json "LMX_FUSE_ROPE=0" | head -2 | cut -c1-420
