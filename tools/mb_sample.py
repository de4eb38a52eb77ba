"""Time the device sampler and argmax at V = 32000 (bf16 logits)."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C
dev = torch.device("cuda:0")
V = 32000
logits = (torch.randn(V, device=dev) * 3).bfloat16()
out = torch.zeros(1, dtype=torch.long, device=dev)
off = torch.zeros(1, dtype=torch.int32, device=dev)
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
for name, (T, p, k) in {"T0.7_p0.9": (0.7, 0.9, 0), "T1_p1": (1.0, 1.0, 0), "T0.8_p0.95_k40": (0.8, 0.95, 40)}.items():
    res[name] = t(lambda: _C.check(_C.lib.lmx_op_sample(_C.DTYPE_BF16, _C.ptr(logits), V, T, p, k, 7, _C.ptr(off), None, _C.ptr(out), None, _C.stream_handle())))
res["argmax"] = t(lambda: _C.check(_C.lib.lmx_op_argmax(_C.DTYPE_BF16, _C.ptr(logits), V, _C.ptr(out), _C.stream_handle())))
print(json.dumps({k: round(v, 2) for k, v in res.items()}))
