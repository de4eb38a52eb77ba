#!/bin/bash
# Round 4, session I: decode weight prefetch beside the attention launch (second stream) — A/B over the budget per layer.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  env "$@" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_i_bench.json 2> gpurun_out/r04_i_bench.err || tail -5 gpurun_out/r04_i_bench.err
  python tools/bench_brief.py gpurun_out/r04_i_bench.json "$*" | sed -n '1p'
}
run LMX_DECODE_PREFETCH=0
run LMX_DECODE_PREFETCH=32
run LMX_DECODE_PREFETCH=64
run LMX_DECODE_PREFETCH=128
run LMX_DECODE_PREFETCH=32 LMX_DECODE_PREFETCH_NT=1
run LMX_DECODE_PREFETCH=32 LMX_DECODE_PREFETCH_BLOCKS=256
run LMX_DECODE_PREFETCH=0
run LMX_DECODE_PREFETCH=64 LMX_DECODE_PREFETCH_BLOCKS=32
