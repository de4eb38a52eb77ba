#!/bin/bash
# Round 5, session E: in-workgroup split-K forms of the 64x64 ring kernel (variants 16 / 17) — parity, then the CLIP tower's four 577-row shapes against the shipping picks
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "gemm_plain or transpose_detecting or gemm_bias_act" 2>&1 | tail -3 ) 2>&1
timeout 200 python tools/mb_gemm_variants.py "577,1024,1024;577,1024,4096;577,3072,1024;577,4096,1024" "15,16,17,5" 5 2>/dev/null | tee gpurun_out/r05_vis_gemm_ksplit.jsonl
