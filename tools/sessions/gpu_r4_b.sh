#!/bin/bash
# Round 4, session B: tile-shaped fused RMSNorm in the split-K reduction (parity + in-situ A/B).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -k "fused_rmsnorm" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_b_norm.log; tail -8 gpurun_out/r04_b_norm.log
for mode in 2 0 2 0; do
  LMX_FUSE_NORM=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_b_bench_norm$mode.json 2> gpurun_out/r04_b_bench.err || tail -5 gpurun_out/r04_b_bench.err
  python tools/bench_brief.py gpurun_out/r04_b_bench_norm$mode.json "FUSE_NORM=$mode"
done
