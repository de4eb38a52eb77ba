#!/bin/bash
# Round-3 GPU session O: the unmodified reference worker (sourceless byte code from oracle/_ref) on the GPU.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
ls oracle/_ref/llava_pyc/llava/serve/ 2>&1 | head -3
timeout 900 python -m pytest tests/test_worker_flow_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.txt
