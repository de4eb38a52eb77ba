"""One weight-gradient GEMM shape for the PMC passes (tools/gpu_pmc.sh): lmx_op_gemm_wgrad = gemm8t_kernel.  Usage: python tools/mb_wgrad_one.py rows out in iters"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    from llava_mi355x import ops
    rows, O, I, iters = (int(v) for v in sys.argv[1:5])
    dev = torch.device("cuda:0")
    dy = (torch.randn((rows, O), device=dev) * 0.1).to(torch.bfloat16)
    x = (torch.randn((rows, I), device=dev) * 0.1).to(torch.bfloat16)
    out = torch.empty((O, I), dtype=torch.bfloat16, device=dev)
    for _ in range(iters):
        ops.gemm_wgrad(dy, x, out)
    torch.cuda.synchronize()
    print("ok", float(out.float().abs().max()))


if __name__ == "__main__":
    main()
