"""Decode-batch linear with / without the RMSNorm of its input rows inside the launch (csrc/skinny.hip XN): interleaved medians over rotating weight copies
(4 x 100 - 260 MB: past the 256 MB memory-side cache, so the weights stream from HBM as in a decode step).  usage: mb_skinny_norm.py [M ...]
Forms of the fused kernel: LMX_SKINNY_XNORM=1 (default: 142 VGPRs, one workgroup per CU) / 2 (held to 128 VGPRs, 52 B of scratch per lane) — one per process."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [2, 8, 16]
    dev = torch.device("cuda")
    dt = torch.bfloat16
    shapes = [("qkv", 12288, 4096, _C.ACT_NONE), ("gate_up", 22016, 4096, _C.ACT_SILU_MUL), ("lm_head", 32000, 4096, _C.ACT_NONE),
              ("qkv_tp8", 1536, 4096, _C.ACT_NONE), ("gate_up_tp8", 2752, 4096, _C.ACT_SILU_MUL)]
    form = os.environ.get("LMX_SKINNY_XNORM", "1")
    for name, N, K, act in shapes:
        n_copy = max(4, int(math.ceil(600e6 / (N * K * 2))))
        ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).to(dt) for _ in range(n_copy)]
        g = (1.0 + 0.1 * torch.randn(K, device=dev)).to(dt)
        for M in Ms:
            x = torch.randn(M, K, device=dev).to(dt)
            xn = torch.empty_like(x)
            out = torch.empty((M, N // 2 if act == _C.ACT_SILU_MUL else N), dtype=dt, device=dev)

            def unfused(w):
                _C.check(_C.lib.lmx_op_rmsnorm(_C.torch_dtype_code(dt), _C.ptr(x), _C.ptr(g), _C.ptr(xn), M, K, 1e-5, _C.stream_handle()))
                ops.gemm(xn, w, act=act, variant=22, out=out)

            def fused(w):
                ops.skinny_gemm_norm(x, w, g, 1e-5, act=act, variant=22, out=out)

            def plain(w):                       # the linear alone on a pre-normalised input: what the fused launch would cost if the norm were free
                ops.gemm(xn, w, act=act, variant=22, out=out)

            res = {}
            for w in ws:                         # build the cached fragment-order copies
                unfused(w); fused(w)
            torch.cuda.synchronize()
            for rnd in range(5):
                for key, fn in (("unfused", unfused), ("fused", fused), ("plain", plain)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(16):
                        fn(ws[i % n_copy])
                    e1.record(); torch.cuda.synchronize()
                    res.setdefault(key, []).append(e0.elapsed_time(e1) / 16 * 1e3)
            med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
            print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "xnorm_form": form, "us": {k: round(v, 2) for k, v in med.items()},
                              "fused_saves_us": round(med["unfused"] - med["fused"], 2)}), flush=True)
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
