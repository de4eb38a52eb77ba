#!/bin/bash
# Round 4, session E: row-major slabs + row-owning fused RMSNorm (LMX_FUSE_NORM=3), scratch-free K-sliced kernels.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_e_gemm8p.log; tail -6 gpurun_out/r04_e_gemm8p.log
for mode in 3 0 3 0; do
  LMX_FUSE_NORM=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_e_bench_norm$mode.json 2> gpurun_out/r04_e_bench.err || tail -5 gpurun_out/r04_e_bench.err
  python tools/bench_brief.py gpurun_out/r04_e_bench_norm$mode.json "FUSE_NORM=$mode" | head -2
done
LMX_SPLITK_MODE=7 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_e_bench_mode7.json 2> gpurun_out/r04_e_bench.err || tail -5 gpurun_out/r04_e_bench.err
python tools/bench_brief.py gpurun_out/r04_e_bench_mode7.json "SPLITK_MODE=7 (row-major slabs, no fused norm)" | head -2
