#!/bin/bash
# Round-3 GPU session R: RoPE + KV append in the q|k|v GEMM epilogue: tests, then the bench with and without.
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qkv_fuse_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py tests/test_full_depth_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.txt
for e in 0 1 0 1; do
LMX_FUSE_ROPE=$e timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-pmc --no-tp-projection > $O/bench_fuse$e.json 2> $O/bench_fuse$e.err; echo "bench fuse=$e rc=$?"
python - $O/bench_fuse$e.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    print({k: round(r[k], 3) for k in ('value','prefill_ms','decode_ms_per_token')}, round(r['roofline_prefill']['frac'], 4))
    kb = r['kernel_breakdown_ms_per_step']
    print({k: (round(v['ms'], 3), v['n']) for k, v in kb.items() if k.startswith('prefill')})
PY
done
