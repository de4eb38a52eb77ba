#!/bin/bash
# Round 4, session N: small-tile epilogues with bias / residual requested ahead of the stores (parity + bench).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_vision_fuse_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py tests/test_gemm8p_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r04_n_tests.log; tail -3 gpurun_out/r04_n_tests.log
for i in 1 2 3; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_n_bench.json 2> gpurun_out/r04_n_bench.err || tail -5 gpurun_out/r04_n_bench.err
  python tools/bench_brief.py gpurun_out/r04_n_bench.json "epilogue preload" | head -2 | cut -c1-560
done
