#!/bin/bash
# One GPU-box session: per-kernel parity (separate processes so a fault in one group cannot hide the others), then microbenchmarks.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
for grp in "gemm_plain or transpose" "bias_act or silu" "gemv" "norms or rope or argmax or im2col" "flash" "decode_attn"; do
  tag=$(echo "$grp" | tr ' ' '_')
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "$grp" -p no:cacheprovider 2>&1 | tail -60 > "gpurun_out/ops_${tag}.log"
  echo "== $grp: $(tail -1 gpurun_out/ops_${tag}.log)"
done
timeout 600 python tools/microbench.py gemm gemv attn > gpurun_out/microbench.log 2>&1
tail -5 gpurun_out/microbench.log
