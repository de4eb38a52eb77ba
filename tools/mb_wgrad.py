"""Weight-gradient GEMM at the 7B training shapes (16 x 2048 = 32768 rows): lmx_op_gemm_wgrad (csrc/gemm8t.hip, operands in their forward layout) against the path it
replaces (two lmx_op_transpose + lmx_op_gemm).  Usage: python tools/mb_wgrad.py [rows]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from llava_mi355x import ops
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    dev = torch.device("cuda:0")
    for name, O, I in (("q|k|v", 12288, 4096), ("o_proj", 4096, 4096), ("gate / up", 11008, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)):
        dy = (torch.randn((rows, O), device=dev) * 0.1).to(torch.bfloat16)
        x = (torch.randn((rows, I), device=dev) * 0.1).to(torch.bfloat16)
        out = torch.empty((O, I), dtype=torch.bfloat16, device=dev)
        out2 = torch.empty_like(out)
        fl = 2.0 * rows * O * I
        us_direct = timed(lambda: ops.gemm_wgrad(dy, x, out))
        us_tr = timed(lambda: (ops.transpose_padded(dy, 64), ops.transpose_padded(x, 64)))
        dyt, xt = ops.transpose_padded(dy, 64), ops.transpose_padded(x, 64)
        us_nt = timed(lambda: ops.gemm(dyt, xt, out=out2))
        print(json.dumps({"linear": name, "rows": rows, "out": O, "in": I, "direct_us": round(us_direct, 1), "direct_TFs": round(fl / us_direct / 1e6, 1),
                          "transposes_us": round(us_tr, 1), "nt_gemm_us": round(us_nt, 1), "nt_gemm_TFs": round(fl / us_nt / 1e6, 1),
                          "old_path_us": round(us_tr + us_nt, 1), "speedup": round((us_tr + us_nt) / us_direct, 3), "bit_identical": bool(torch.equal(out, out2))}))
        del dy, x, out, out2, dyt, xt


if __name__ == "__main__":
    main()
