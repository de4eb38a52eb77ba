#!/bin/bash
# Round 5, session O (r5-I): flash prefill with per-key-group arrival counters instead of the workgroup barrier, group 1 staggered
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
rm -f gpurun_out/r05_flash_gsync.jsonl
for cfg in "0 0" "1 0" "1 1" "1 2" "1 3" "1 4" "1 6" "1 8"; do
  set -- $cfg
  LMX_FLASH_GSYNC=$1 LMX_FLASH_STAGGER=$2 timeout 120 python tools/mb_flash_sync.py 40 2>/dev/null | tee -a gpurun_out/r05_flash_gsync.jsonl
done
LMX_FLASH_GSYNC=1 LMX_FLASH_STAGGER=2 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -p no:cacheprovider -k "flash or forward_matches or vision" 2>&1 | tail -3
