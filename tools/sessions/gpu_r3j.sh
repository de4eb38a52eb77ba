#!/bin/bash
# Round-3 GPU session J: gemv2m timeline (probes), P = 8.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_attn_merge_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for c in "" "LMX_ATTN_PROBE=1" "LMX_ATTN_MERGE=0"; do
  timeout 300 python tools/mb_decode.py "$c" --tokens 64 2>&1 | tail -1
done | tee $O/mb_decode_probe.txt
