#!/bin/bash
# Round 4, session C: fused RMSNorm (granule exchange) A/B + rank-local TP shapes over the GEMM variants.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -q -k "fused_rmsnorm" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04_c_norm.log; tail -4 gpurun_out/r04_c_norm.log
for mode in 2 0 2 0; do
  LMX_FUSE_NORM=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch --no-tp-projection --no-pmc > gpurun_out/r04_c_bench_norm$mode.json 2> gpurun_out/r04_c_bench.err || tail -5 gpurun_out/r04_c_bench.err
  python tools/bench_brief.py gpurun_out/r04_c_bench_norm$mode.json "FUSE_NORM=$mode" | head -2
done
timeout 900 python tools/mb_gemm_variants.py "1087,6144,4096;1087,4096,2048;1087,11008,4096;1087,4096,5504;1087,3072,4096;1087,4096,1024;1087,5504,4096;1087,4096,2752;1087,1536,4096;1087,4096,512;1087,2752,4096;1087,4096,1408" "0,1,4,7,9,14,15,18,30,33,34,-1" 3 > gpurun_out/r04_c_tp_gemm.jsonl 2>&1
python - <<'P'
import json
best = {}
for l in open("gpurun_out/r04_c_tp_gemm.jsonl"):
    try: r = json.loads(l)
    except Exception: continue
    k = (r["M"], r["N"], r["K"])
    if "us_median" in r: best.setdefault(k, []).append((r["us_median"], r["variant"]))
for k, v in best.items():
    v.sort(); print(k, "auto", [x for x in v if x[1] == 0], "best", v[:4], "blaslt", [x for x in v if x[1] == -1])
P
