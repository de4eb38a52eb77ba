#!/bin/bash
# Round-2 GPU session K: (1) GEMV-after-touch microbenchmark, (2) config 4 harness under TP=2 (two ranks sharing the box's GPU).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2k; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== gemv prefetch probe"
timeout 300 python tools/mb_gemv_prefetch.py 2>&1 | grep gemv_prefetch | tee $O/gemv_prefetch.jsonl
echo "== config4 TP=2 shared GPU"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/config4_harness.py \
    --model llava_plus_v0_13b --requests 32 --batch 32 --shared-gpu > $O/config4_tp2.log 2> $O/config4_tp2.err
grep '^{' $O/config4_tp2.log | cut -c1-1500
tail -5 $O/config4_tp2.err | cut -c1-400
