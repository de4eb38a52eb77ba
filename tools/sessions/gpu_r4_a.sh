#!/bin/bash
# Round 4, session A: LDS-DMA landing probe (two-phase K-step decision), the new parity closures, vision-shape GEMM variant sweep.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/probes/ldsdma_landing > gpurun_out/r04_ldsdma_landing.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r04_ldsdma_landing.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemv2_real or decode_attn_flow" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_a_ops.log; tail -5 gpurun_out/r04_a_ops.log
timeout 600 python -m pytest tests/test_stop_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_a_stop.log; tail -5 gpurun_out/r04_a_stop.log
timeout 900 python -m pytest tests/test_full_depth_gpu.py -m gpu -q -k "fp32_engine_full_depth" -p no:cacheprovider -s 2>&1 | tail -15 > gpurun_out/r04_a_fd.log; tail -6 gpurun_out/r04_a_fd.log
timeout 600 python tools/mb_gemm_variants.py "577,3072,1024;577,1024,1024;577,4096,1024;577,1024,4096" "0,1,4,5,7,14,15,18,-1" 5 > gpurun_out/r04_a_vis_gemm.jsonl 2>&1; tail -40 gpurun_out/r04_a_vis_gemm.jsonl
