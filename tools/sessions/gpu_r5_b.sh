#!/bin/bash
# Round 5, session B: the pruned library + the split-q decode step (decode_attn.hip) — op-level and engine-level bit-identity, then the in-situ A/B.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== op level"
( timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "decode_kv_attn or decode_attn_step" 2>&1 | tail -15 ) 2>&1
echo "== engine level"
( timeout 400 python -m pytest tests/test_decode_splitq_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -15 ) 2>&1
for m in 0 1; do
  echo "== LMX_DECODE_SPLITQ=$m"
  LMX_DECODE_SPLITQ=$m timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > gpurun_out/r05_bench_splitq$m.json 2>> gpurun_out/r05_b.err
  python tools/bench_brief.py gpurun_out/r05_bench_splitq$m.json "LMX_DECODE_SPLITQ=$m" | head -4
done
tail -5 gpurun_out/r05_b.err
