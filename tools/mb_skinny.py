"""Skinny (decode-batch) linear microbenchmark: GB/s of the weight stream per shape and M, weights rotated through 6 buffers."""
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops

VARIANT = int(os.environ.get("SKINNY_VARIANT", "20"))      # 22 = fragment-order weight copy (cached per weight)


def main():
    dev = torch.device("cuda:0")
    shapes = [("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate_up", 22016, 4096, 3), ("down", 4096, 11008, 0), ("lm_head", 32000, 4096, 0)]
    Ms = [int(x) for x in os.environ.get("MS", "1,8,32").split(",")]
    for name, N, K, act in shapes:
        ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(6)]
        for M in Ms:
            x = torch.randn(M, K, device=dev).bfloat16()
            out = torch.empty((M, N // 2 if act == 3 else N), dtype=torch.bfloat16, device=dev)
            for w in ws[:2]:
                ops.gemm(x, w, act=act, variant=VARIANT, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 18
            e0.record()
            for r in range(reps):
                ops.gemm(x, ws[r % 6], act=act, variant=VARIANT, out=out)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            row = {"shape": name, "N": N, "K": K, "M": M, "us": round(us, 2), "GBps": round(N * K * 2 / us / 1e3, 1)}
            if M == 1 and act != 3:
                e0.record()
                for r in range(reps):
                    ops.gemv(x, ws[r % 6], out=out)
                e1.record(); torch.cuda.synchronize()
                row["gemv_us"] = round(e0.elapsed_time(e1) / reps * 1e3, 2)
            print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
