#!/bin/bash
# Round-2 GPU session I: split-K publish mode 4 (plain stores, sc1 loads) vs the shipping mode 1: stress + in-model timing.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r2i; mkdir -p $O
export TMPDIR=/tmp
for mode in 1 4; do
echo "== split-K mode $mode: stress"
LMX_SPLITK_MODE=$mode timeout 300 python - 2>&1 <<'PY' | tail -3
import math, sys, torch
sys.path.insert(0, "llava-plus-codebase_amd")
from llava_mi355x import ops
dev = torch.device("cuda:0")
bad = 0; n = 0
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device=dev)
for shape in ((1087, 4096, 11008), (1087, 4096, 4096), (2000, 5120, 13824), (513, 776, 2048)):
    M, N, K = shape
    for it in range(40):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        ref = ops.gemm(x, w, variant=35)
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk2 = junk @ junk
        got = ops.gemm(x, w, variant=34 if it % 2 else 33)
        torch.cuda.synchronize()
        d = (got.float() - ref.float()).abs().max().item(); n += 1
        if d > 2.0 ** -6 * ref.float().abs().max().item(): bad += 1
print("mode mismatches", bad, "of", n)
PY
for rep in 1 2; do
  LMX_SPLITK_MODE=$mode timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-batch --no-pmc > $O/bench_m${mode}_$rep.json 2> $O/bench_m${mode}_$rep.err
  python - $O/bench_m${mode}_$rep.json $mode <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    kb = r['kernel_breakdown_ms_per_step']
    print('mode=%s' % sys.argv[2], {k: round(r[k], 3) for k in ('value','prefill_ms')}, 'frac', round(r['roofline_prefill']['frac'], 4),
          {k.split('.')[-1]: round(kb[k]['ms'] / kb[k]['n'] * 1e3, 1) for k in kb if k.startswith('prefill.gemm')}, 'ids_same', r['greedy_ids_identical_across_steps'])
PY
done; done
