#!/bin/bash
# Round 4, session F: CLIP K/V^T pack in the q|k|v epilogue (parity + A/B), deeper-ring GEMM arms on the CLIP and TP-rank shapes, unloaded round-2 TTFT of config 4.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vision_fuse_gpu.py tests/test_model_gpu.py tests/test_real_geometry_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_f_tests.log; tail -6 gpurun_out/r04_f_tests.log
for v in 1 0 1 0; do
  LMX_VIS_PACK=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_f_bench_pack$v.json 2> gpurun_out/r04_f_bench.err || tail -5 gpurun_out/r04_f_bench.err
  python tools/bench_brief.py gpurun_out/r04_f_bench_pack$v.json "VIS_PACK=$v" | head -2
done
timeout 900 python tools/mb_gemm_variants.py "577,3072,1024;577,1024,1024;577,4096,1024;577,1024,4096;1087,1536,4096;1087,4096,512;1087,2752,4096;1087,4096,1408;1087,3072,4096;1087,4096,1024;1087,5504,4096;1087,4096,2752" "0,5,14,15,18,21,22,23,24,-1" 5 > gpurun_out/r04_f_gemm.jsonl 2>&1
python - <<'P'
import json
best = {}
for l in open("gpurun_out/r04_f_gemm.jsonl"):
    try: r = json.loads(l)
    except Exception: continue
    k = (r["M"], r["N"], r["K"])
    if "us_median" in r: best.setdefault(k, []).append((r["us_median"], r["variant"]))
    elif "error" in r: print("ERR", r)
for k, v in best.items():
    v.sort(); print(k, "auto", [x for x in v if x[1] == 0], "best", v[:4])
P
for r in 0 1; do
  timeout 600 python tools/config4_harness.py --model llava_plus_v0_13b --requests 1 --batch 1 --reuse $r > gpurun_out/r04_f_config4_single_reuse$r.json 2> gpurun_out/r04_f_config4.err || tail -20 gpurun_out/r04_f_config4.err
  python -c "
import json
for l in open('gpurun_out/r04_f_config4_single_reuse$r.json'):
    try: d = json.loads(l)
    except Exception: continue
    print('single reuse=$r', {k: d.get(k) for k in ('completed','median_ttft_s','median_round2_ttft_s','median_total_s','answers_as_scripted','reuse')})
"
done
