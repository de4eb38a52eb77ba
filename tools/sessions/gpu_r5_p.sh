#!/bin/bash
# Round 5, session P: the live PMC pass of the prefill GEMM shapes inside bench.py (kernel-name filter fixed after the prune)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python - <<'PY'
import json, os, sys
sys.argv = ["bench.py"]
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
r = b.pmc_prefill_traffic(os.getcwd(), 1087)
print(json.dumps(r))
H, I, T, es = 4096, 11008, 1087, 2
algo = {"qkv": (T * H + 3 * H * H + T * 3 * H) * es, "o_proj": (T * H + H * H + 2 * T * H) * es, "gate_up": (T * H + 2 * H * I + T * I) * es, "down": (T * I + H * I + 2 * T * H) * es}
if r and "error" not in r: print({k: round(r[k] / algo[k], 2) for k in algo})
PY
