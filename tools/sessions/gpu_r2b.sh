#!/bin/bash
# Round-2 GPU session B: ping-pong GEMM arms, full-depth parity (7B, 13B, fp32 cut), headline bench.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
echo "== gemm8p tests"; timeout 600 python -m pytest tests/test_gemm8p_gpu.py -x -q > $O/test_gemm8p.log 2>&1; echo "rc=$?"; tail -5 $O/test_gemm8p.log
echo "== microbench"; timeout 600 python tools/mb_gemm_variants.py "1087,12288,4096;1087,22016,4096;1087,4096,11008;4096,4096,4096;8192,8192,8192" "9,30,35,36,37,38,39,40,-1,30,36,37,38" > $O/mb_gemm.jsonl 2>&1; echo "rc=$?"; cat $O/mb_gemm.jsonl
echo "== bench"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 800 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2b/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline_prefill', {k: r['roofline_prefill'][k] for k in ('achieved','frac','avg_launch_us','prefill_end_to_end_frac')})
    for k, v in r['kernel_breakdown_ms_per_step'].items():
        if k.startswith('prefill') or k.startswith('vis') or k.startswith('proj'): print(f'  {k:26s} {v["ms"]:10.3f} ms  n={v["n"]}')
PY
echo "== full depth"; timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -s > $O/test_full_depth.log 2>&1; echo "rc=$?"; grep -E "passed|failed|Error|fp32_8layer" $O/test_full_depth.log | tail -12
cp gpurun_out/full_depth_*.json $O/ 2>/dev/null
echo "== model + real geometry tests"; timeout 900 python -m pytest tests/test_model_gpu.py tests/test_real_geometry_gpu.py tests/test_loader_gpu.py tests/test_worker_flow_gpu.py tests/test_batching_gpu.py -x -q > $O/test_model.log 2>&1; echo "rc=$?"; tail -5 $O/test_model.log
