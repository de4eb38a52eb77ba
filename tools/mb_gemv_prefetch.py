"""Does touching a weight matrix shortly before its GEMV pay?  (decode idea: a prefetch kernel running beside the latency-bound attention
kernel pulls the next GEMVs' weights toward L2 / Infinity Cache.)  Cycle over 32 distinct matrices per shape as real decode does; variant
'touch' reads matrix i+1 with a plain streaming-sum kernel (torch) right before the GEMV on matrix i... so by the time GEMV i+1 runs its
weights were read once ~one kernel earlier.  Reported: GEMV-only time from events around each GEMV (event pairs on both variants)."""
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C, ops
dev = torch.device("cuda:0")
L = 16
for name, N, K in [("o_proj", 4096, 4096), ("down", 4096, 11008), ("qkv", 12288, 4096), ("gate_up", 22016, 4096)]:
    ws = [(torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16() for _ in range(L)]
    x = torch.randn(1, K, device=dev).bfloat16(); g = torch.ones(K, device=dev).bfloat16()
    act = _C.ACT_SILU_MUL if name == "gate_up" else _C.ACT_NONE
    out = torch.empty(1, N // 2 if act else N, device=dev, dtype=torch.bfloat16)
    nw = g if name in ("qkv", "gate_up") else None
    sink = torch.zeros(1, device=dev, dtype=torch.int32)
    res = {}
    for variant in ("cold", "touch", "cold", "touch"):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(L)]
        for rep in range(3):
            for i, w in enumerate(ws):
                if variant == "touch":
                    nxt = ws[(i + 1) % L]
                    sink += nxt.view(torch.int32).sum(dtype=torch.int32)          # streaming read of the NEXT matrix
                evs[i][0].record()
                ops.gemv(x, w, norm_w=nw, act=act, out=out)
                evs[i][1].record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)[L // 2]
        res.setdefault(variant, []).append(round(us, 2))
    print(json.dumps({"kind": "gemv_prefetch", "name": name, "MB": round(N * K * 2 / 1e6, 1), "median_us": res}), flush=True)
    del ws
