"""Per-kernel microbenchmarks at LLaVA-1.5-7B shapes (run on the GPU box).  Writes JSON lines to gpurun_out/."""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops  # noqa: E402

dev = torch.device("cuda:0")
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
results = []


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rec(**kw):
    results.append(kw)
    print(json.dumps(kw), flush=True)


def bench_gemm():
    shapes = [("qkv", 1087, 12288, 4096), ("o_proj", 1087, 4096, 4096), ("gate_up", 1087, 22016, 4096), ("down", 1087, 4096, 11008),
              ("clip_qkv", 577, 3072, 1024), ("clip_fc1", 577, 4096, 1024), ("clip_fc2", 577, 1024, 4096), ("sq4096", 4096, 4096, 4096)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        for variant in (4, 5, 7, 9, 14, 15):
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            try:
                t = timeit(lambda: ops.gemm(x, w, variant=variant, out=out))
            except Exception as e:  # noqa: BLE001
                rec(kind="gemm", name=name, variant=variant, error=str(e)); continue
            rec(kind="gemm", name=name, M=M, N=N, K=K, variant=variant, us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12)
        # torch (hipBLASLt) as an on-box yardstick only — never on the product path
        t = timeit(lambda: torch.matmul(x, w.t()))
        rec(kind="gemm", name=name, M=M, N=N, K=K, variant="torch_hipblaslt_yardstick", us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12)


def bench_gemv():
    for name, N, K in [("qkv", 12288, 4096), ("o_proj", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]:
        x = torch.randn(1, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        g = torch.ones(K, device=dev).bfloat16()
        act = _C.ACT_SILU_MUL if name == "gate_up" else _C.ACT_NONE
        out = torch.empty(1, N // 2 if act else N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemv(x, w, norm_w=g if name in ("qkv", "gate_up", "lm_head") else None, act=act, out=out), iters=50)
        rec(kind="gemv", name=name, N=N, K=K, us=t * 1e6, gbps=N * K * 2 / t / 1e9)


def bench_attn():
    D, nh = 128, 32
    for T in (1087, 2048):
        q = torch.randn(T, nh * D, device=dev).bfloat16()
        kc, vt = ops.alloc_kv(nh, 2048 + 64, D, torch.bfloat16)
        kc.normal_(); vt.normal_()
        out = torch.empty(T, nh * D, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, True, out=out))
        rec(kind="flash_prefill", T=T, us=t * 1e6, tflops=2.0 * T * T * nh * D / t / 1e12)
    # CLIP shape
    D, nh, T = 64, 16, 577
    q = torch.randn(T, nh * D, device=dev).bfloat16()
    kc, vt = ops.alloc_kv(nh, 640, D, torch.bfloat16); kc.normal_(); vt.normal_()
    out = torch.empty(T, nh * D, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, False, out=out))
    rec(kind="flash_clip", T=T, us=t * 1e6, tflops=4.0 * T * T * nh * D / t / 1e12)
    # decode attention at ctx 1100
    D, nh = 128, 32
    q = torch.randn(1, nh * D, device=dev).bfloat16()
    kc, vt = ops.alloc_kv(nh, 4096, D, torch.bfloat16); kc.normal_(); vt.normal_()
    for ns in (4, 8, 16):
        t = timeit(lambda: ops.decode_attn(q, kc, vt, 1, 1100, 0, nh, nh, D, True, n_split=ns), iters=50)
        rec(kind="decode_attn", ctx=1101, n_split=ns, us=t * 1e6, gbps=2 * 1101 * nh * D * 2 / t / 1e9)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "gemv", "attn"]
    t0 = time.time()
    for w in which:
        {"gemm": bench_gemm, "gemv": bench_gemv, "attn": bench_attn}[w]()
    with open(os.path.join(OUT, "microbench.jsonl"), "a") as f:
        for r in results:
            f.write(json.dumps(r) + "\n")
    print("done in %.1fs" % (time.time() - t0))
