#!/bin/bash
# Round-3 GPU session P: beam search with stopping criteria / image batches.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_beam_gpu.py tests/test_api_surface_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.txt
