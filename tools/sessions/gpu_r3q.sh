#!/bin/bash
# Round-3 GPU session Q: 13B widths, TP = 2 (hooked), packed prefill, bf16 vs oracle.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_real_geometry_gpu.py -x -q -m gpu -k "13b_tp2" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.txt
