"""Run ONE GEMM variant/shape a few times (for rocprofv3 --pmc passes)."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops
variant, M, N, K = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(x, w, variant=variant, out=out)
torch.cuda.synchronize()
