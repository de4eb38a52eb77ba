"""Single-op wrappers over the C ABI (`lmx_op_*`): the same HIP kernels the engine launches, exposed for unit parity
tests and microbenchmarks.  torch tensors are containers only (pointers + current stream)."""
from __future__ import annotations

import math

import torch

from . import _C
from ._C import check, lib, ptr, stream_handle, torch_dtype_code


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("llava_mi355x ops need tensors on the MI355X (cuda) device; there is no CPU fallback")
        if t is not None and not t.is_contiguous():
            raise ValueError("llava_mi355x ops need contiguous tensors")


def gemm(x, w, bias=None, residual=None, act=_C.ACT_NONE, variant=0, out=None):
    """act(x @ w.T + bias) (+ residual).  x [M,K], w [N,K]; SiLU·mul expects w as the fused [32 gate|32 up] layout."""
    _need_cuda(x, w, bias, residual)
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if act == _C.ACT_SILU_MUL else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    check(lib.lmx_op_gemm(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(out), ptr(bias), ptr(residual), M, N, K,
                          x.stride(0), w.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0,
                          act, variant, stream_handle()), "gemm")
    return out


def gemv(x, w, bias=None, residual=None, norm_w=None, eps=1e-5, act=_C.ACT_NONE, out=None):
    """Decode-time linear for 1..4 rows: act(norm(x) @ w.T + bias) (+ residual)."""
    _need_cuda(x, w, bias, residual, norm_w)
    MB, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if act == _C.ACT_SILU_MUL else N
    if out is None:
        out = torch.empty((MB, n_out), dtype=x.dtype, device=x.device)
    check(lib.lmx_op_gemv(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(out), ptr(bias), ptr(residual), ptr(norm_w), eps,
                          MB, N, K, x.stride(0), w.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0,
                          act, stream_handle()), "gemv")
    return out


def rmsnorm(x, w, eps):
    _need_cuda(x, w)
    y = torch.empty_like(x)
    check(lib.lmx_op_rmsnorm(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(y), x.shape[0], x.shape[1], eps, stream_handle()), "rmsnorm")
    return y


def layernorm(x, w, b, eps):
    _need_cuda(x, w, b)
    y = torch.empty_like(x)
    check(lib.lmx_op_layernorm(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(b), ptr(y), x.shape[0], x.shape[1], eps, stream_handle()), "layernorm")
    return y


def interleave_gate_up(gate, up):
    """Host-side statement of the fused [32 gate | 32 up] row layout the SiLU·mul epilogue expects (tests only;
    the engine does this on the device at weight-load time)."""
    I, K = gate.shape
    assert I % 32 == 0
    g = gate.view(I // 32, 32, K)
    u = up.view(I // 32, 32, K)
    return torch.cat([g, u], dim=1).reshape(2 * I, K).contiguous()


def alloc_kv(n_kv_heads, s_max, head_dim, dtype, device="cuda"):
    """Zero-initialised K cache [kvh][s_max][D] and Vᵀ cache [kvh][D][s_max] (one layer)."""
    k = torch.zeros((n_kv_heads, s_max, head_dim), dtype=dtype, device=device)
    vt = torch.zeros((n_kv_heads, head_dim, s_max), dtype=dtype, device=device)
    return k, vt


def rope_kv(qkv, kcache, vtcache, cos_sin, pos0, n_heads, n_kv_heads, head_dim):
    """Rotate q in place, write rotated k rows and v columns into the caches at positions pos0.."""
    _need_cuda(qkv, kcache, vtcache, cos_sin)
    T = qkv.shape[0]
    assert qkv.shape[1] == (n_heads + 2 * n_kv_heads) * head_dim
    s_max = kcache.shape[1]
    check(lib.lmx_op_rope_kv(torch_dtype_code(qkv.dtype), head_dim, ptr(qkv), ptr(kcache), ptr(vtcache), ptr(cos_sin), pos0, T,
                             n_heads, n_kv_heads, s_max, stream_handle()), "rope_kv")


def flash_attn(q, kcache, vtcache, q_len, kv_len, q_pos0, n_heads, n_kv_heads, head_dim, causal, q_stride=None, out=None):
    """Prefill attention over the caches.  q: [q_len, q_stride] with head h at column h*D."""
    _need_cuda(q, kcache, vtcache)
    if q_stride is None:
        q_stride = q.stride(0)
    if out is None:
        out = torch.empty((q_len, n_heads * head_dim), dtype=q.dtype, device=q.device)
    s_max = kcache.shape[1]
    check(lib.lmx_op_flash_attn(torch_dtype_code(q.dtype), head_dim, ptr(q), ptr(out), ptr(kcache), ptr(vtcache), q_len, kv_len, q_pos0,
                                q_stride, out.stride(0), n_heads, n_kv_heads, s_max, 1.0 / math.sqrt(head_dim), int(causal),
                                stream_handle()), "flash_attn")
    return out


def decode_attn(q, kcache, vtcache, n_rows, pos0, kv_total, n_heads, n_kv_heads, head_dim, causal, n_split=1, q_stride=None, out=None):
    _need_cuda(q, kcache, vtcache)
    if q_stride is None:
        q_stride = q.stride(0)
    if out is None:
        out = torch.empty((n_rows, n_heads * head_dim), dtype=q.dtype, device=q.device)
    s_max = kcache.shape[1]
    ws = torch.empty(lib.lmx_op_decode_attn_ws_bytes(n_rows, n_heads, n_split, head_dim) // 4, dtype=torch.float32, device=q.device)
    check(lib.lmx_op_decode_attn(torch_dtype_code(q.dtype), head_dim, ptr(q), ptr(out), ptr(kcache), ptr(vtcache), n_rows, pos0, kv_total,
                                 int(causal), q_stride, out.stride(0), n_heads, n_kv_heads, s_max, n_split, 1.0 / math.sqrt(head_dim),
                                 ptr(ws), stream_handle()), "decode_attn")
    return out


def argmax(logits):
    _need_cuda(logits)
    out = torch.zeros(1, dtype=torch.int64, device=logits.device)
    check(lib.lmx_op_argmax(torch_dtype_code(logits.dtype), ptr(logits), logits.numel(), ptr(out), stream_handle()), "argmax")
    return out


def im2col(pixels, patch, kpad):
    _need_cuda(pixels)
    N, C, S, _ = pixels.shape
    P = (S // patch) ** 2
    out = torch.empty((N * P, kpad), dtype=pixels.dtype, device=pixels.device)
    check(lib.lmx_op_im2col(torch_dtype_code(pixels.dtype), ptr(pixels), ptr(out), N, S, patch, kpad, stream_handle()), "im2col")
    return out
