"""Standalone timing of decode_fused_kernel over 32 layers of KV cache (ctx ~1100), with debug modes to attribute time."""
import math, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C
from llava_mi355x._C import lib, ptr, stream_handle, check
dev = torch.device("cuda:0")
D, nh, s_max, L = 128, 32, 2048, 32
qkv = torch.randn(3 * nh * D, device=dev).bfloat16()
kc = torch.randn(L, nh, s_max, D, device=dev).bfloat16()
vt = torch.randn(L, nh, D, s_max, device=dev).bfloat16()
cs = torch.randn(s_max, D, device=dev)
ns = s_max // 128
ws = torch.zeros(nh * ns * (D + 4), device=dev)
cnt = torch.zeros(nh, dtype=torch.int32, device=dev)
out = torch.empty(nh * D, device=dev, dtype=torch.bfloat16)
for ctx in (1100, 300):
    pos = torch.tensor([ctx], dtype=torch.int32, device=dev)
    for mode in (0, 1, 2):
        def run():
            for l in range(L):
                check(lib.lmx_op_decode_fused(_C.DTYPE_BF16, D, ptr(qkv), ptr(kc[l]), ptr(vt[l]), ptr(cs), ptr(pos), nh, nh, s_max, 1 / math.sqrt(D),
                                              ptr(ws), ptr(cnt), ptr(out), mode, stream_handle()))
        for _ in range(3): run()
        cnt.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (10 * L) * 1e3
        print(json.dumps({"kind": "decode_fused", "ctx": ctx, "debug_mode": mode, "us_per_launch": us, "gbps": 2 * (ctx + 1) * nh * D * 2 / us / 1e3}), flush=True)
        cnt.zero_()
# empty kernel launch rate reference: argmax on tiny
x = torch.randn(64, device=dev).bfloat16(); o = torch.zeros(1, dtype=torch.int64, device=dev)
def run2():
    for _ in range(32): check(lib.lmx_op_argmax(_C.DTYPE_BF16, ptr(x), 64, ptr(o), stream_handle()))
for _ in range(3): run2()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run2()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"kind": "tiny_kernel_back_to_back", "us": e0.elapsed_time(e1) / 320 * 1e3}))
