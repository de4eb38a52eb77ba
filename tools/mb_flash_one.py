"""Run the prefill flash-attention kernel a few times at the 7B prompt shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1087
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D, nh = 128, 32
dev = torch.device("cuda:0")
q = torch.randn(T, nh * D, device=dev).bfloat16()
kc, vt = ops.alloc_kv(nh, 2048 + 64, D, torch.bfloat16)
kc.normal_(); vt.normal_()
out = torch.empty(T, nh * D, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, True, out=out)
torch.cuda.synchronize()
