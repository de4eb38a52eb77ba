#!/bin/bash
# Round-2 GPU session C: final ping-pong GEMM form + split-K write-through, API-surface tests, full suite, bench.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
echo "== gemm8p + api tests"; timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_api_surface_gpu.py tests/test_loader_gpu.py -q > $O/test_a.log 2>&1; echo "rc=$?"; tail -25 $O/test_a.log
echo "== microbench"; timeout 600 python tools/mb_gemm_variants.py "1087,12288,4096;1087,22016,4096;1087,4096,4096;1087,4096,11008;4096,4096,4096" "30,31,32,33,34,35,18,-1" 5 > $O/mb_gemm.jsonl 2>&1; echo "rc=$?"; cat $O/mb_gemm.jsonl
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 800 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2c/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline_prefill', {k: r['roofline_prefill'][k] for k in ('achieved','frac','avg_launch_us','prefill_end_to_end_frac')})
    for k, v in r['kernel_breakdown_ms_per_step'].items():
        if k.startswith('prefill'): print(f'  {k:26s} {v["ms"]:10.3f} ms  n={v["n"]}')
PY
echo "== full depth"; timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q > $O/test_full_depth.log 2>&1; echo "rc=$?"; tail -5 $O/test_full_depth.log
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_full_depth_gpu.py --deselect tests/test_gemm8p_gpu.py --deselect tests/test_api_surface_gpu.py --deselect tests/test_loader_gpu.py > $O/test_gpu.log 2>&1; echo "rc=$?"; tail -8 $O/test_gpu.log
