"""Time GEMM variants on given shapes (weights rotated through 4 buffers). Usage: mb_gemm_variants.py "M,N,K;M,N,K" "v1,v2,..." """
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1].split(";")]
variants = [int(v) for v in sys.argv[2].split(",")]
dev = torch.device("cuda:0")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(4)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = (x.float() @ ws[0].float().t())
    for v in variants:
        try:
            ops.gemm(x, ws[0], variant=v, out=out)
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            for r in range(3):
                ops.gemm(x, ws[r % 4], variant=v, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for r in range(reps):
                ops.gemm(x, ws[r % 4], variant=v, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            print(json.dumps({"M": M, "N": N, "K": K, "variant": v, "us": round(us, 1), "TFs": round(2.0 * M * N * K / us / 1e6, 1), "rel_err": round(err, 5)}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"M": M, "N": N, "K": K, "variant": v, "error": str(e)[:200]}), flush=True)
