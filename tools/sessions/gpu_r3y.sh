#!/bin/bash
# Round-3 GPU session Y: (rounds in flight, rows per wave) sweep of gemv2_kernel after the counted x / norm / residual loads — per-launch table per shape.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3y; mkdir -p $O
export TMPDIR=/tmp
for c in "LMX_GEMV2=2,1;" "LMX_GEMV2=4,1;" "LMX_GEMV2=8,1;" "LMX_GEMV2=2,2;" "LMX_GEMV2=4,2;" "LMX_GEMV2=8,2;" "LMX_GEMV2=2,4;" "LMX_GEMV2=4,4;"; do
  timeout 200 python tools/mb_decode.py "$c" --tokens 48 --rounds 2 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); u=r['us_per_launch']
print(r['config'].ljust(16), r['best_us'], {k.split('.')[-1]: u[k] for k in u if 'gemv' in k})"
done | tee $O/sweep.txt
