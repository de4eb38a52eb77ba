#!/bin/bash
# Round 4, session s: two-shot all-reduce with G fat workgroups (flags per workgroup): real-size message between two processes on one GPU, G = 16 .. 544.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 544 128 64 32 16 8; do
  echo "== LMX_P2P_BIG_WGS=$g" | tee -a gpurun_out/r04_p2p_big_wgs.txt
  ( LMX_P2P_BIG_WGS=$g timeout 200 python -m pytest tests/test_tp_p2p_gpu.py -q -x -s -k real_width -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep "us_per\|passed\|failed\|Error" | tail -4 ) 2>&1 | tee -a gpurun_out/r04_p2p_big_wgs.txt
done
( time timeout 600 python -m pytest tests/test_tp_p2p_gpu.py tests/test_tp_serving_gpu.py tests/test_tp_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3 ) 2>&1
