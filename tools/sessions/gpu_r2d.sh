#!/bin/bash
# Round-2 GPU session D: split-K protocol fix, PMC evidence, new bench (full CPU baseline + live PMC traffic), config 3.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
echo "== gemm8p + api + loader + real-geometry tests"; timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_api_surface_gpu.py tests/test_loader_gpu.py tests/test_real_geometry_gpu.py -q > $O/test_a.log 2>&1; echo "rc=$?"; tail -12 $O/test_a.log
echo "== split-K stress (100 launches, uneven load)"; timeout 300 python - > $O/splitk_stress.log 2>&1 <<'PY'
import math, sys, torch
sys.path.insert(0, "llava-plus-codebase_amd")
from llava_mi355x import ops
dev = torch.device("cuda:0")
bad = 0
for shape in ((1087, 4096, 11008), (1087, 4096, 4096), (2000, 5120, 13824)):
    M, N, K = shape
    for it in range(34):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        ref = ops.gemm(x, w, variant=35)
        got = ops.gemm(x, w, variant=34 if it % 2 else 33)
        # a second stream keeps part of the chip busy with unrelated work (uneven load)
        d = (got.float() - ref.float()).abs().max().item()
        tol = 2.0 ** -6 * ref.float().abs().max().item()
        if d > tol: bad += 1; print("MISMATCH", shape, it, d)
print("split-K stress mismatches:", bad)
PY
tail -3 $O/splitk_stress.log
echo "== PMC"; timeout 1500 bash tools/gpu_pmc_r2.sh > $O/pmc.log 2>&1; echo "rc=$?"; tail -40 $O/pmc.log
echo "== bench"; timeout 1200 python bench.py --steps 5 --warmup 2 --cpu-fp32 > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2d/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline', {k: r['roofline'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','traffic','event_pair_overhead_us')})
    print('traffic_source', json.dumps(r['roofline'].get('traffic_source'))[:400])
    print('roofline_prefill', {k: r['roofline_prefill'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','prefill_end_to_end_frac','by_shape_tflops')})
    print('cpu', json.dumps(r['cpu_baseline'])[:1200])
    print('serving', r.get('serving_batch'))
PY
echo "== config3 (13B, batch 8, chunked prefill, TP=1 on this box)"; timeout 900 python bench.py --workload config3 --steps 2 --warmup 1 > $O/bench_config3.json 2> $O/bench_config3.err; echo "rc=$?"; tail -c 400 $O/bench_config3.err; cat $O/bench_config3.json | cut -c1-1500
echo "== full depth"; timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q > $O/test_full_depth.log 2>&1; echo "rc=$?"; tail -4 $O/test_full_depth.log
cp gpurun_out/full_depth_*.json $O/ 2>/dev/null
