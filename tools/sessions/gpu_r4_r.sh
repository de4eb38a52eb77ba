#!/bin/bash
# Round 4, session r: why the two-shot all-reduce of a real-size message takes 349 us on one GPU — copies through uncached / fine-grained memory (probe), the
# same all-reduce with fine-grained exchange buffers; the fused-norm policy in place (batching tests).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/uncached_bw.hip -o /tmp/uncached_bw 2>/dev/null && timeout 120 /tmp/uncached_bw | tee gpurun_out/r04_uncached_bw.txt
for mem in uncached finegrained; do
  echo "== exchange buffers: $mem"
  ( LMX_P2P_MEM=$mem timeout 300 python -m pytest tests/test_tp_p2p_gpu.py -q -x -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -8 ) 2>&1 | tee -a gpurun_out/r04_p2p_mem_ab.txt
done
( time timeout 600 python -m pytest tests/test_batching_gpu.py tests/test_tp_serving_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3 ) 2>&1
