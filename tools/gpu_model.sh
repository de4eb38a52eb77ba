#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/model_tests.log
tail -40 gpurun_out/model_tests.log
