#!/bin/bash
# Round 5, session O (r5-I): flash prefill — the compiler's vmcnt waits on the Q fragments inside the key loop also waited for the next round's DMAs
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for cfg in "0 0"; do
  set -- $cfg
  LMX_FLASH_GSYNC=$1 LMX_FLASH_STAGGER=$2 timeout 120 python tools/mb_flash_sync.py 40 2>/dev/null | tee -a gpurun_out/r05_flash_gsync.jsonl
done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py -m gpu -q -x -p no:cacheprovider -k "flash or forward_matches or vision or one_layer" 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-tp-projection --no-batch > gpurun_out/r05_o_bench.json 2>/dev/null; python tools/bench_brief.py gpurun_out/r05_o_bench.json "after the wait fix" | head -3
