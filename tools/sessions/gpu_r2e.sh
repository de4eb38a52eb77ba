#!/bin/bash
# Round-2 GPU session E: split-K publish-protocol experiment, tool-loop test, TP 2-process tests, bench with calibrated overhead.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2e; mkdir -p $O
export TMPDIR=/tmp
for mode in 0 1 2; do
echo "== split-K mode $mode: stress + timing"
LMX_SPLITK_MODE=$mode timeout 300 python - > $O/splitk_mode$mode.log 2>&1 <<'PY'
import math, sys, time, torch
sys.path.insert(0, "llava-plus-codebase_amd")
from llava_mi355x import ops
dev = torch.device("cuda:0")
bad = 0; n = 0
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
for shape in ((1087, 4096, 11008), (1087, 4096, 4096), (2000, 5120, 13824), (513, 776, 2048)):
    M, N, K = shape
    for it in range(24):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        ref = ops.gemm(x, w, variant=35)
        if it % 3 == 0:
            with torch.cuda.stream(side):           # uneven load: an unrelated kernel shares the chip
                junk2 = junk @ junk
        got = ops.gemm(x, w, variant=34 if it % 2 else 33)
        torch.cuda.synchronize()
        d = (got.float() - ref.float()).abs().max().item(); n += 1
        if d > 2.0 ** -6 * ref.float().abs().max().item(): bad += 1
print("mismatches", bad, "of", n)
for M, N, K, v in ((1087, 4096, 11008, 34), (1087, 4096, 4096, 33)):
    x = torch.randn(M, K, device=dev).bfloat16(); ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(4)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10): ops.gemm(x, ws[i % 4], variant=v, out=out)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    print(M, N, K, "variant", v, "us median", sorted(ts)[2])
PY
cat $O/splitk_mode$mode.log | tail -4
done
echo "== tests: gemm8p (mode 0), tool loop, worker flow, tp p2p, tp gpu"; timeout 1200 python -m pytest tests/test_gemm8p_gpu.py tests/test_tool_loop_gpu.py tests/test_worker_flow_gpu.py tests/test_tp_p2p_gpu.py tests/test_tp_gpu.py tests/test_real_geometry_gpu.py -q > $O/test_a.log 2>&1; echo "rc=$?"; tail -12 $O/test_a.log
echo "== bench"; timeout 1200 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -c 400 $O/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r2e/bench.json"):
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in ('value','ms_per_step','prefill_ms','decode_tokens_per_s','decode_ms_per_token','greedy_ids_identical_across_steps')})
    print('roofline', {k: r['roofline'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','traffic','event_pair_overhead_us','event_pair_calibration')})
    print('roofline_prefill', {k: r['roofline_prefill'].get(k) for k in ('achieved','frac','frac_with_event_overhead','avg_launch_us','prefill_end_to_end_frac','by_shape_tflops')})
PY
echo "== config4 harness (13B LLaVA-Plus geometry, 32 concurrent tool-loop requests)"; timeout 900 python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 --batch 32 > $O/config4.json 2> $O/config4.err; echo "rc=$?"; tail -c 600 $O/config4.err; cut -c1-1500 $O/config4.json
