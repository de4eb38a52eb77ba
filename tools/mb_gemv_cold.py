"""Cold-cache GEMV timing: cycle over 32 distinct weight matrices per shape (no L2 / Infinity-Cache reuse), like real decode."""
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops
dev = torch.device("cuda:0")
L = 32
for name, N, K in [("qkv", 12288, 4096), ("o_proj", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]:
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(L)]
    x = torch.randn(1, K, device=dev).bfloat16(); g = torch.ones(K, device=dev).bfloat16()
    act = _C.ACT_SILU_MUL if name == "gate_up" else _C.ACT_NONE
    out = torch.empty(1, N // 2 if act else N, device=dev, dtype=torch.bfloat16)
    nw = g if name in ("qkv", "gate_up") else None
    def run():
        for w in ws:
            ops.gemv(x, w, norm_w=nw, act=act, out=out)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * L) * 1e3
    print(json.dumps({"kind": "gemv_cold", "name": name, "us": round(us, 2), "gbps": round(N * K * 2 / us / 1e3, 1)}), flush=True)
    del ws
