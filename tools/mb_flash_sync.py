"""Time the prefill flash-attention launch at the 7B prompt shape (T = 1087, 32 heads x 128, causal) and at CLIP's (577 tokens, 16 heads x 64, not causal),
rotating over several caches, and print a checksum of the output (the forms under test must agree bit for bit).  The kernel's experiment switches are read
from the environment once per process: run one process per setting.   usage: mb_flash_sync.py [iters]"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
out = {"gsync": os.environ.get("LMX_FLASH_GSYNC", "0"), "stagger": os.environ.get("LMX_FLASH_STAGGER", "-")}
for name, T, nh, D, causal in (("llm_1087", 1087, 32, 128, True), ("llm_2048", 2048, 32, 128, True), ("clip_577", 577, 16, 64, False)):
    torch.manual_seed(0)
    q = torch.randn(T, nh * D, device=dev).bfloat16()
    caches = []
    for i in range(6):
        kc, vt = ops.alloc_kv(nh, 2048 + 64, D, torch.bfloat16)
        kc.normal_(); vt.normal_()
        caches.append((kc, vt))
    o = torch.empty(T, nh * D, device=dev, dtype=torch.bfloat16)
    for kc, vt in caches:
        ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, causal, out=o)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            kc, vt = caches[i % len(caches)]
            ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, causal, out=o)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    kc, vt = caches[0]
    ops.flash_attn(q, kc, vt, T, T, 0, nh, nh, D, causal, out=o)
    torch.cuda.synchronize()
    out[name] = {"us": round(sorted(ts)[2], 2), "sha": hashlib.sha1(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:10]}
print(json.dumps(out), flush=True)
