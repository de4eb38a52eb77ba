#!/bin/bash
# Round-3 GPU session M: row-order split-K reduction.
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for e in 1 0; do
LMX_SPLITK_ROWS=$e timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $O/bench_rows$e.json 2> $O/bench_rows$e.err; echo "bench rows=$e rc=$?"
python - $O/bench_rows$e.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_ms_per_token')}, r['roofline_prefill']['frac'], r['roofline_prefill']['by_shape_tflops'])
    kb = r['kernel_breakdown_ms_per_step']
    print({k: (round(v['ms'], 3), v['n']) for k, v in kb.items() if k.startswith('prefill')})
PY
done
