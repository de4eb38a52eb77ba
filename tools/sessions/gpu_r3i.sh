#!/bin/bash
# Round-3 GPU session I: chunk merge moved into o_proj's staging (gemv2m_kernel) + counted x / RMSNorm-weight / residual loads in gemv2_kernel.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_merge_gpu.py tests/test_ops_gpu.py tests/test_decode_flow_gpu.py tests/test_decode_engine_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for c in "" "LMX_ATTN_MERGE=0" "LMX_ATTN_MERGE=0;LMX_GEMV2=0"; do
  timeout 300 python tools/mb_decode.py "$c" --tokens 64 2>&1 | tail -4
done | tee $O/mb_decode.txt
