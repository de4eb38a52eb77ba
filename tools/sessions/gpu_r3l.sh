#!/bin/bash
# Round-3 GPU session L: flash prefill with the permuted key order (one 16-byte V^T read per P.V fragment): tests, time, LDS conflict counters.
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py tests/test_full_depth_gpu.py tests/test_batching_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"
python - $O/bench_short.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_ms_per_token')})
    kb = r['kernel_breakdown_ms_per_step']
    print({k: (round(v['ms'], 3), v['n']) for k, v in kb.items() if k.startswith('prefill') or k.startswith('vis')})
PY
SHAPES=flash bash tools/gpu_pmc_r3.sh 2>&1 | tail -30
