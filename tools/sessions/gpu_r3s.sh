#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_qkv_fuse_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_decode_flow_gpu.py tests/test_decode_engine_gpu.py tests/test_real_geometry_gpu.py -q -x -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "^E  |PASS|FAIL|passed|failed" $O/pytest.txt | tail -24
