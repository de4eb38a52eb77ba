#!/bin/bash
# register-resident RMSNorm / LayerNorm kernels: bit-identity test, then the prefill in situ with the switch off and on
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3norm; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_ops_gpu.py -q -k "norms" 2>&1 | tail -3 | tee $O/pytest.txt
for v in 0 1 0 1; do
  LMX_NORM_REG=$v timeout 100 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection 2>/dev/null | tail -1 > $O/b$v.json
  python - <<PY
import json
d=json.load(open("$O/b$v.json")); k=d["kernel_breakdown_ms_per_step"]
print("NORM_REG=$v value", round(d["value"],1), "prefill_ms", round(d["prefill_ms"],3), "rmsnorm", k.get("prefill.rmsnorm"), "vis.layernorm", k.get("vis.layernorm"))
PY
done | tee $O/ab.txt
