#!/bin/bash
# Full GPU-box session: all gpu tests, then bench (+ optional rocprof).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/tests_gpu.log; tail -5 gpurun_out/tests_gpu.log
PROFILE=${PROFILE:-0} bash tools/gpu_bench.sh
