"""Time GEMM variants on given shapes.  Usage: mb_gemm_variants.py "M,N,K;M,N,K" "v1,v2,..." [rounds]
Variants are timed in interleaved rounds inside ONE process (guide §5.4 rule 24: a perf delta needs a within-probe A/B; single runs on
this box move by several % with the DVFS state the previous kernel left behind); median and min over the rounds are reported.
Weights rotate through 4 buffers (no L2-warm weight panel); variant < 0 = hipBLASLt through torch, an on-box yardstick only."""
import json, math, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1].split(";")]
variants = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    NB = int(os.environ.get('MB_NBUF', '4'))         # 1: the same weight every repetition (Infinity-Cache-warm when N K 2 B < 256 MB)
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(NB)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = (x.float() @ ws[0].float().t())
    fns, errs, times = {}, {}, {v: [] for v in variants}
    for v in variants:
        fns[v] = (lambda w_: torch.matmul(x, w_.t(), out=out)) if v < 0 else (lambda w_, v=v: ops.gemm(x, w_, variant=v, out=out))
        try:
            fns[v](ws[0])
            errs[v] = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            for r in range(3):
                fns[v](ws[r % NB])
        except Exception as e:  # noqa: BLE001
            errs[v] = str(e)[:200]
    torch.cuda.synchronize()
    reps = 10
    for _ in range(rounds):
        for v in variants:
            if isinstance(errs[v], str):
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(reps):
                fns[v](ws[r % NB])
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps * 1e3)
    for v in variants:
        if isinstance(errs[v], str):
            print(json.dumps({"M": M, "N": N, "K": K, "variant": v, "error": errs[v]}), flush=True); continue
        med, mn = statistics.median(times[v]), min(times[v])
        print(json.dumps({"M": M, "N": N, "K": K, "variant": v, "us_median": round(med, 1), "us_min": round(mn, 1), "TFs_median": round(2.0 * M * N * K / med / 1e6, 1),
                          "TFs_best": round(2.0 * M * N * K / mn / 1e6, 1), "rel_err": round(errs[v], 5), "rounds": rounds}), flush=True)
