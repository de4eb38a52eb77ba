"""The four decoder linears of the headline prefill (LLaVA-1.5-7B, 1087 rows), as the engine launches them, a few times each in ONE process — the driver of
bench.py's in-run `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` passes (roofline_prefill.traffic).  Order: q|k|v, gate|up (variant 0: un-split ping-pong kernel),
o_proj, down_proj (variant 30: K-sliced ping-pong + launch-boundary reduction — the single-op entry brings no split-K scratch, so variant 0 would not slice).
Usage: mb_gemm_prefill.py [iters] [T]"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1087
H, I = 4096, 11008
dev = torch.device("cuda:0")
torch.manual_seed(0)
def W(n, k): return (torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16()
x = torch.randn(T, H, device=dev).bfloat16(); act = torch.randn(T, I, device=dev).bfloat16(); res = torch.randn(T, H, device=dev).bfloat16()
wqkv, wo, wgu, wd = W(3 * H, H), W(H, H), W(2 * I, H), W(H, I)
for _ in range(iters): ops.gemm(x, wqkv, variant=0)
torch.cuda.synchronize()
for _ in range(iters): ops.gemm(x, wgu, act=_C.ACT_SILU_MUL, variant=0)
torch.cuda.synchronize()
for _ in range(iters): ops.gemm(x, wo, residual=res, variant=30)
torch.cuda.synchronize()
for _ in range(iters): ops.gemm(act, wd, residual=res, variant=30)
torch.cuda.synchronize()
