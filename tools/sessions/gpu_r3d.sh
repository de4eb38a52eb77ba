#!/bin/bash
# Round-3 GPU session D: N = hidden prefill GEMMs (o_proj, down_proj): launch-boundary split-K reduction vs the in-launch reduction, + the GEMM tests.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
SH="1087,4096,4096;1087,4096,11008;1087,5120,5120;1087,5120,13824"
for mode in 5 1; do
  echo "== LMX_SPLITK_MODE=$mode"
  LMX_SPLITK_MODE=$mode timeout 300 python tools/mb_gemm_variants.py "$SH" "30,33,34,18,-1" 5 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r['M'], r['N'], r['K'], 'v', r['variant'], r.get('us_median'), 'us', r.get('TFs_median'), 'TF', r.get('rel_err', r.get('error')))
" | tee $O/gemm_mode$mode.txt
done
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -x -q 2>&1 | tail -3
