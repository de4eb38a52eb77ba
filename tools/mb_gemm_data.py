"""Does the DATA set the GEMM's rate?  The ping-pong kernel (gemm8p) on one large shape with operands that are all zero, a constant, small-range random and N(0, 1):
the instruction stream is identical, only the operand toggling (= MFMA power) differs.  Usage: python tools/mb_gemm_data.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    from llava_mi355x import ops
    dev = torch.device("cuda:0")
    M, N, K = 16384, 12288, 4096
    fl = 2.0 * M * N * K
    kinds = {
        "zeros": lambda s: torch.zeros(s, device=dev),
        "ones": lambda s: torch.ones(s, device=dev),
        "randn": lambda s: torch.randn(s, device=dev),
        "randn x 1e-3": lambda s: torch.randn(s, device=dev) * 1e-3,
        "sign only (+-1)": lambda s: torch.sign(torch.randn(s, device=dev)),
        "one mantissa pattern (randn rounded to powers of two)": lambda s: torch.sign(torch.randn(s, device=dev)) * torch.exp2(torch.round(torch.randn(s, device=dev))),
    }
    for name, mk in kinds.items():
        x = mk((M, K)).to(torch.bfloat16); w = mk((N, K)).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm(x, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.gemm(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(json.dumps({"operands": name, "M": M, "N": N, "K": K, "us": round(us, 1), "TFs": round(fl / us / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
