#!/bin/bash
# Round 4, session J: gate|up tile quantisation, third attempt — cheap ragged tiles + tail-split with a launch-boundary reduction.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r04_j_gemm8p.log; tail -5 gpurun_out/r04_j_gemm8p.log
timeout 900 python tools/mb_gemm_variants.py "1087,22016,4096;1087,12288,4096;1087,27648,5120;2048,22016,4096" "35,36,-1" 7 > gpurun_out/r04_j_gemm.jsonl 2>&1; cat gpurun_out/r04_j_gemm.jsonl | cut -c1-200
LMX_GEMM8P_TAIL_INLAUNCH=1 timeout 600 python tools/mb_gemm_variants.py "1087,22016,4096" "35,36" 5 2>&1 | cut -c1-200
for v in 1 0 1 0; do
  LMX_GEMM8P_TAIL=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_j_bench_tail$v.json 2> gpurun_out/r04_j_bench.err || tail -5 gpurun_out/r04_j_bench.err
  python tools/bench_brief.py gpurun_out/r04_j_bench_tail$v.json "GEMM8P_TAIL=$v" | head -2
done
