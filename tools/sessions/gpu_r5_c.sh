#!/bin/bash
# Round 5, session C: rows per wave / rounds in flight of the k|v projection inside the split-q launch (grid = 288 attention + N / (4 R) projection workgroups over 1024 slots)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for rp in 24 44 42 28 18; do
  echo "== LMX_KVATTN_RP=$rp"
  ( LMX_KVATTN_RP=$rp timeout 200 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "decode_kv_attn" 2>&1 | tail -1 ) 2>&1
  LMX_KVATTN_RP=$rp timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > gpurun_out/r05_bench_kvattn_rp$rp.json 2>> gpurun_out/r05_c.err
  python tools/bench_brief.py gpurun_out/r05_bench_kvattn_rp$rp.json "rp=$rp" | grep -v "prefill ms"
done
