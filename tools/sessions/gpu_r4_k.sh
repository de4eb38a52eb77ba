#!/bin/bash
# Round 4, session K: why is the tail-split order slow?  All-sliced (33: S = 2, 34: S = 3) vs plain (35) vs tail split (36) on the big shapes.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/mb_gemm_variants.py "1087,12288,4096;1087,22016,4096;1024,22016,4096;1024,8192,4096" "35,33,34,36" 5 2>&1 | cut -c1-160 | tee gpurun_out/r04_k_gemm.jsonl
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profk -o run -- python $OLDPWD/tools/mb_gemm_variants.py "1087,22016,4096" "35,36" 2 > /dev/null 2>&1; cd $OLDPWD
f=$(find /tmp/profk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_k_kernel_stats.csv && head -8 "$f" | cut -c1-200
