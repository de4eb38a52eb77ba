"""The training attention backward of one sample for the PMC passes (tools/gpu_pmc.sh): flash forward with lse, then lmx_op_attn_bwd_lse (attn_delta / attn_bwd_dq / attn_bwd_dkv
kernels), 32 heads x 128.  Usage: python tools/mb_attn_bwd_one.py T iters"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    from llava_mi355x import ops
    T, iters = int(sys.argv[1]), int(sys.argv[2])
    nh = nkv = 32; D = 128
    dev = torch.device("cuda:0")
    qkv = (torch.randn((T, (nh + 2 * nkv) * D), device=dev) * 0.5).to(torch.bfloat16)
    kc, vt = ops.alloc_kv(nkv, -(-T // 128) * 128, D, torch.bfloat16, dev)
    table = torch.cat([torch.ones(T, D // 2), torch.zeros(T, D // 2)], -1).to(dev)
    ops.rope_kv(qkv, kc, vt, table, 0, nh, nkv, D, k_rows=True)
    lse = torch.zeros((nh, -(-T // 64) * 64), dtype=torch.float32, device=dev)
    out = ops.flash_attn(qkv, kc, vt, T, T, 0, nh, nkv, D, True, lse=lse)
    d_out = (torch.randn((T, nh * D), device=dev) * 0.5).to(torch.bfloat16)
    q = qkv[:, :nh * D].contiguous(); k = qkv[:, nh * D:(nh + nkv) * D].contiguous(); v = qkv[:, (nh + nkv) * D:].contiguous()
    for _ in range(iters):
        dq, dk, dv = ops.attn_bwd_lse(q, k, v, out, d_out, lse, nh, nkv, D)
    torch.cuda.synchronize()
    print("ok", float(dq.float().abs().max()))


if __name__ == "__main__":
    main()
