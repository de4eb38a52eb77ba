#!/bin/bash
# Round-3 GPU session N: device-side stop rule.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_stop_gpu.py tests/test_batching_gpu.py tests/test_sampling_gpu.py tests/test_api_surface_gpu.py tests/test_worker_flow_gpu.py tests/test_tool_loop_gpu.py tests/test_beam_gpu.py tests/test_tp_serving_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
