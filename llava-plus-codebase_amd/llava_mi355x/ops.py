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


def _need_cuda_rows(*ts):
    """As _need_cuda for 2-D operands whose ROWS are contiguous but whose row stride is free (column windows of wider buffers): the entry point takes the stride."""
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("llava_mi355x ops need tensors on the MI355X (cuda) device; there is no CPU fallback")
        if t is not None and (t.dim() != 2 or t.stride(1) != 1):
            raise ValueError("llava_mi355x ops need 2-D operands with contiguous rows")


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


def rope_kv(qkv, kcache, vtcache, cos_sin, pos0, n_heads, n_kv_heads, head_dim, k_rows=False):
    """Rotate q in place, write rotated k rows and v columns into the caches at positions pos0..; k_rows: the rotated k also replaces the k columns of qkv."""
    _need_cuda(qkv, kcache, vtcache, cos_sin)
    T = qkv.shape[0]
    assert qkv.shape[1] == (n_heads + 2 * n_kv_heads) * head_dim
    s_max = kcache.shape[1]
    fn = lib.lmx_op_rope_kv_rows if k_rows else lib.lmx_op_rope_kv
    check(fn(torch_dtype_code(qkv.dtype), head_dim, ptr(qkv), ptr(kcache), ptr(vtcache), ptr(cos_sin), pos0, T, n_heads, n_kv_heads, s_max, stream_handle()), "rope_kv")


def gemm_qkv_rope(x, w, qkv, kcache, vtcache, cos_sin, pos0, n_heads, n_kv_heads, head_dim):
    """q|k|v projection with RoPE + KV append in the GEMM epilogue: rotated q into qkv[:, :n_heads*D], rotated k / v into the caches at pos0.."""
    _need_cuda(x, w, qkv, kcache, vtcache, cos_sin)
    T, K = x.shape
    assert w.shape == ((n_heads + 2 * n_kv_heads) * head_dim, K) and qkv.shape == (T, w.shape[0]) and x.is_contiguous() and w.is_contiguous() and qkv.is_contiguous()
    check(lib.lmx_op_gemm_qkv_rope(torch_dtype_code(x.dtype), head_dim, ptr(x), ptr(w), ptr(qkv), ptr(kcache), ptr(vtcache), ptr(cos_sin), pos0, T, K,
                                   n_heads, n_kv_heads, kcache.shape[1], stream_handle()), "gemm_qkv_rope")


def flash_attn(q, kcache, vtcache, q_len, kv_len, q_pos0, n_heads, n_kv_heads, head_dim, causal, q_stride=None, out=None, lse=None):
    """Prefill attention over the caches.  q: [q_len, q_stride] with head h at column h*D.  lse (optional, [n_heads, >= q_len] float32): receives the log2-domain
    log-sum-exp of every row's scaled scores — what attn_bwd_lse takes instead of recomputing it."""
    _need_cuda(q, kcache, vtcache, lse)
    if q_stride is None:
        q_stride = q.stride(0)
    if out is None:
        out = torch.empty((q_len, n_heads * head_dim), dtype=q.dtype, device=q.device)
    s_max = kcache.shape[1]
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.shape[0] == n_heads and lse.shape[1] >= q_len
        check(lib.lmx_op_flash_attn_lse(torch_dtype_code(q.dtype), head_dim, ptr(q), ptr(out), ptr(kcache), ptr(vtcache), q_len, kv_len, q_pos0, q_stride, out.stride(0),
                                        n_heads, n_kv_heads, s_max, 1.0 / math.sqrt(head_dim), int(causal), ptr(lse), lse.stride(0), stream_handle()), "flash_attn_lse")
        return out
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


def decode_attn_scratch(n_heads, n_split, head_dim, dtype, device):
    """(partials workspace, zeroed arrival counters, output row) of the decode-step attention launches; reusable across launches (the merger re-arms the counters)"""
    return (torch.zeros(n_heads * n_split * (head_dim + 4), dtype=torch.float32, device=device), torch.zeros(n_heads, dtype=torch.int32, device=device),
            torch.empty(n_heads * head_dim, dtype=dtype, device=device))


def decode_attn_step(qkv, kcache, vtcache, cos_sin, pos, n_heads, n_kv_heads, head_dim, scratch=None):
    """The decode step's attention launch (16-bit models): RoPE(q, k_new) at `pos`, K / V^T append, attention over keys 0..pos.  qkv: one [q|k|v] row
    (pre-RoPE).  Returns the [n_heads * head_dim] output row."""
    _need_cuda(qkv, kcache, vtcache, cos_sin)
    s_max = kcache.shape[1]
    n_split = (s_max + 127) // 128
    ws, cnt, out = scratch if scratch is not None else decode_attn_scratch(n_heads, n_split, head_dim, qkv.dtype, qkv.device)
    check(lib.lmx_op_decode_attn_step(torch_dtype_code(qkv.dtype), head_dim, ptr(qkv), ptr(kcache), ptr(vtcache), ptr(cos_sin), int(pos), n_heads, n_kv_heads,
                                      s_max, 1.0 / (head_dim ** 0.5), ptr(ws), ptr(cnt), ptr(out), stream_handle()), "decode_attn_step")
    return out


def decode_attn_batch(qkv, kcaches, vtcaches, pos_words, cos_sin, n_heads, n_kv_heads, head_dim, n_split=None):
    """The decode BATCH's attention launch (continuous batching): row z of qkv [n_seq, (n_heads + 2 n_kv_heads) head_dim] (pre-RoPE) against sequence z's own
    caches kcaches[z] [n_kv_heads, s_max, D] / vtcaches[z] [n_kv_heads, D, s_max] and its device position word pos_words[z] (int32 [1]: keys already cached).
    Appends the rotated key / the value at that position and returns the [n_seq, n_heads * head_dim] attention rows."""
    import ctypes
    n_seq = qkv.shape[0]
    _need_cuda(qkv, cos_sin, *kcaches, *vtcaches, *pos_words)
    assert len(kcaches) == len(vtcaches) == len(pos_words) == n_seq and qkv.stride(1) == 1
    s_max = kcaches[0].shape[1]
    if n_split is None:
        n_split = (s_max + 127) // 128
    arr = lambda ts: (ctypes.c_void_p * n_seq)(*[ptr(t) for t in ts])
    ws = [torch.zeros(n_heads * n_split * (head_dim + 4), dtype=torch.float32, device=qkv.device) for _ in range(n_seq)]
    cnt = [torch.zeros(n_heads, dtype=torch.int32, device=qkv.device) for _ in range(n_seq)]
    tab = torch.empty(int(lib.lmx_op_decode_attn_batch_tab_bytes(n_seq)), dtype=torch.uint8, device=qkv.device)
    out = torch.empty(n_seq, n_heads * head_dim, dtype=qkv.dtype, device=qkv.device)
    check(lib.lmx_op_decode_attn_batch(torch_dtype_code(qkv.dtype), head_dim, ptr(qkv), qkv.stride(0), arr(kcaches), arr(vtcaches), arr(pos_words), arr(ws), arr(cnt),
                                       n_seq, ptr(cos_sin), n_heads, n_kv_heads, s_max, int(n_split), 1.0 / (head_dim ** 0.5), ptr(tab), ptr(out), out.stride(0),
                                       stream_handle()), "decode_attn_batch")
    return out


def decode_kv_attn(q_row, x, w_kv, norm_w, eps, kcache, vtcache, cos_sin, pos, n_heads, n_kv_heads, head_dim, granules=None, tag=1, timeline=None, scratch=None):
    """The split-q decode step's second launch: the k | v projection of the (RMS-normalised, when norm_w is given) input row x next to the attention
    workgroups.  q_row: a [q|k|v]-sized row whose q columns hold the pre-RoPE query (the k | v columns are ignored).  Same effects as gemv + decode_attn_step."""
    _need_cuda(q_row, x, w_kv, kcache, vtcache, cos_sin)
    s_max = kcache.shape[1]
    n_split = (s_max + 127) // 128
    ws, cnt, out = scratch if scratch is not None else decode_attn_scratch(n_heads, n_split, head_dim, x.dtype, x.device)
    if granules is None:
        granules = torch.zeros(2 * n_kv_heads * head_dim, dtype=torch.int64, device=x.device)
    check(lib.lmx_op_decode_kv_attn(torch_dtype_code(x.dtype), head_dim, ptr(q_row), ptr(x), ptr(w_kv), ptr(norm_w), float(eps), x.shape[-1], w_kv.stride(0),
                                    ptr(kcache), ptr(vtcache), ptr(cos_sin), int(pos), n_heads, n_kv_heads, s_max, 1.0 / (head_dim ** 0.5), ptr(ws), ptr(cnt),
                                    ptr(granules), int(tag), ptr(out), ptr(timeline), stream_handle()), "decode_kv_attn")
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


# ---- training-step slices (csrc/train.hip) -----------------------------------------------------------------------------------------
def ce_loss(logits, labels, ignore_index=-100, want_grad=False, grad=1.0):
    """Shifted cross-entropy of LlamaForCausalLM.forward: logits [B,T,V] (model dtype), labels [B,T] int64 -> (loss fp32 scalar tensor,
    counted positions, dlogits | None).  Mean over positions whose label (at t+1) is not ignore_index; fp32 math on the stored logits."""
    _need_cuda(logits, labels)
    B, T, V = logits.shape
    R = B * (T - 1)
    lse = torch.empty(R, dtype=torch.float32, device=logits.device); row = torch.empty_like(lse)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    d = torch.empty_like(logits) if want_grad else None
    check(lib.lmx_op_ce_loss(torch_dtype_code(logits.dtype), ptr(logits), logits.stride(1), ptr(labels), B, T, V, int(ignore_index), ptr(lse), ptr(row),
                             ptr(out), float(grad), ptr(d), d.stride(1) if d is not None else 0, stream_handle()), "ce_loss")
    return out[0], out[1], d


def rmsnorm_bwd(x, w, dy, eps, want_dw=True, residual=None):
    """Autograd of LlamaRMSNorm: (dx, dw).  residual: the gradient arriving over the skip connection, added to dx in the same pass (dx = residual + dx_norm)."""
    _need_cuda(x, w, dy, residual)
    rows, H = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty(H, dtype=torch.float32, device=x.device) if want_dw else None
    inv = torch.empty((rows + 63) // 64 * 64 + (rows + 127) // 128 * H, dtype=torch.float32, device=x.device)   # inverse norms + the dw partial rows
    if residual is None:
        check(lib.lmx_op_rmsnorm_bwd(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(dy), ptr(dx), ptr(dw), ptr(inv), rows, H, eps, stream_handle()), "rmsnorm_bwd")
    else:
        assert residual.shape == x.shape and residual.dtype == x.dtype
        check(lib.lmx_op_rmsnorm_bwd_add(torch_dtype_code(x.dtype), ptr(x), ptr(w), ptr(dy), ptr(residual), ptr(dx), ptr(dw), ptr(inv), rows, H, eps, stream_handle()),
              "rmsnorm_bwd_add")
    return dx, dw


def swiglu_bwd(gate, up, dact):
    _need_cuda(gate, up, dact)
    dg, du = torch.empty_like(gate), torch.empty_like(up)
    check(lib.lmx_op_swiglu_bwd(torch_dtype_code(gate.dtype), ptr(gate), ptr(up), ptr(dact), ptr(dg), ptr(du), gate.numel(), stream_handle()), "swiglu_bwd")
    return dg, du


def rope_bwd(dy, cos_sin, pos0, heads, head_dim):
    """dy [T, heads*head_dim] (gradient w.r.t. the ROTATED q or k) -> gradient w.r.t. the un-rotated one."""
    _need_cuda(dy, cos_sin)
    dx = torch.empty_like(dy)
    check(lib.lmx_op_rope_bwd(torch_dtype_code(dy.dtype), ptr(dy), ptr(dx), ptr(cos_sin), pos0, dy.shape[0], heads, head_dim, dy.stride(0), stream_handle()), "rope_bwd")
    return dx


def transpose(x):
    _need_cuda(x)
    r, c = x.shape
    out = torch.empty((c, r), dtype=x.dtype, device=x.device)
    check(lib.lmx_op_transpose(torch_dtype_code(x.dtype), ptr(x), x.stride(0), r, c, ptr(out), out.stride(0), stream_handle()), "transpose")
    return out


def linear_bwd(x, w, dy):
    """Gradients of y = x @ w.T on the forward GEMM kernels: dx = dy @ w (contract over N: w transposed once), dw = dy.T @ x (contract over M)."""
    dx = gemm(dy, transpose(w))
    dw = gemm(transpose(dy), transpose(x))
    return dx, dw


def attn_bwd(q, k, v, d_out, heads, kv_heads, head_dim):
    """Causal attention backward.  q, d_out: [T, heads*D]; k, v: [T, kv_heads*D] (k, q already rotated) -> dq, dk, dv."""
    _need_cuda(q, k, v, d_out)
    T = q.shape[0]
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    s1 = torch.empty(T * kv_heads * head_dim, dtype=torch.float32, device=q.device); s2 = torch.empty_like(s1)
    check(lib.lmx_op_attn_bwd(torch_dtype_code(q.dtype), head_dim, ptr(q), ptr(k), ptr(v), ptr(d_out), ptr(dq), ptr(s1), ptr(s2), ptr(dk), ptr(dv), T, heads,
                              kv_heads, q.stride(0), k.stride(0), d_out.stride(0), 1.0 / math.sqrt(head_dim), stream_handle()), "attn_bwd")
    return dq, dk, dv


def attn_bwd_lse(q, k, v, out, d_out, lse, heads, kv_heads, head_dim):
    """attn_bwd with the forward's output and log-sum-exp (flash_attn(..., lse=...)) as inputs: no statistics sweep in the backward (16-bit dtypes)."""
    _need_cuda(q, k, v, out, d_out, lse)
    T = q.shape[0]
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    check(lib.lmx_op_attn_bwd_lse(torch_dtype_code(q.dtype), head_dim, ptr(q), ptr(k), ptr(v), ptr(out), ptr(d_out), ptr(lse), lse.stride(0), ptr(dq), ptr(dk), ptr(dv),
                                  T, heads, kv_heads, q.stride(0), k.stride(0), d_out.stride(0), out.stride(0), 1.0 / math.sqrt(head_dim), stream_handle()), "attn_bwd_lse")
    return dq, dk, dv


# ---- pieces of one optimisation step (csrc/train.hip, composed by llava_mi355x/train.py) ----------------------------------------------
EW_SWIGLU, EW_GELU, EW_GELU_BWD, EW_ADD = 0, 1, 2, 3


def elementwise(op, a, b=None, out=None):
    """op 0: silu(a) * b (HF rounding points); 1: gelu_erf(a); 2: b * gelu'(a); 3: a + b.  Contiguous tensors of one shape."""
    _need_cuda(a, b)
    assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.shape == a.shape))
    if out is None:
        out = torch.empty_like(a)
    check(lib.lmx_op_elementwise(torch_dtype_code(a.dtype), int(op), ptr(a), ptr(b), ptr(out), a.numel(), stream_handle()), "elementwise")
    return out


def cast_f32(src, out):
    """fp32 accumulator -> `out` (parameter dtype), same number of elements."""
    _need_cuda(src, out)
    assert src.dtype == torch.float32 and src.numel() == out.numel() and src.is_contiguous() and out.is_contiguous()
    check(lib.lmx_op_cast_f32(torch_dtype_code(out.dtype), ptr(src), ptr(out), src.numel(), stream_handle()), "cast_f32")
    return out


def col_sum(dy):
    """fp32 column sums of dy [rows, cols] (bias gradient)."""
    _need_cuda(dy)
    out = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
    check(lib.lmx_op_col_sum(torch_dtype_code(dy.dtype), ptr(dy), dy.stride(0), dy.shape[0], dy.shape[1], ptr(out), stream_handle()), "col_sum")
    return out


def gather_embed(src, table, feats=None):
    """Rows of `table` (src >= 0), zero rows (src == -1) and rows of `feats` (src == -2 - k) -> [len(src), H]."""
    _need_cuda(src, table, feats)
    assert src.dtype == torch.int32 and table.is_contiguous() and (feats is None or feats.is_contiguous())
    out = torch.empty((src.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
    check(lib.lmx_op_gather_embed(torch_dtype_code(table.dtype), ptr(src), ptr(table), ptr(feats), ptr(out), src.numel(), table.shape[1], stream_handle()),
          "gather_embed")
    return out


def embed_bwd(src, d_embeds, dtable32=None, dfeats=None):
    """Autograd of gather_embed: token rows add into dtable32 (fp32 [V, H], caller zeroes it), feature rows are copied into dfeats."""
    _need_cuda(src, d_embeds, dtable32, dfeats)
    assert d_embeds.is_contiguous()
    check(lib.lmx_op_embed_bwd(torch_dtype_code(d_embeds.dtype), ptr(src), ptr(d_embeds), ptr(dtable32), ptr(dfeats), src.numel(), d_embeds.shape[1],
                               stream_handle()), "embed_bwd")


def sumsq(x, acc):
    """acc[0] += sum(x^2) (fp32 device scalar)."""
    _need_cuda(x, acc)
    assert x.is_contiguous() and acc.dtype == torch.float32
    check(lib.lmx_op_sumsq(torch_dtype_code(x.dtype), ptr(x), x.numel(), ptr(acc), stream_handle()), "sumsq")


def adamw(param, grad, master, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, gnorm_sq=None, max_grad_norm=0.0):
    """One torch.optim.AdamW update on fp32 master weights; `param` receives the rounded copy.  gnorm_sq: device fp32 scalar holding the
    squared global gradient norm (clip factor min(1, max_grad_norm / (norm + 1e-6)) is applied on the device)."""
    _need_cuda(param, grad, master, exp_avg, exp_avg_sq, gnorm_sq)
    n = param.numel()
    assert grad.numel() == n and master.numel() == n and exp_avg.numel() == n and exp_avg_sq.numel() == n and grad.dtype == param.dtype
    check(lib.lmx_op_adamw(torch_dtype_code(param.dtype), ptr(param), ptr(grad), ptr(master), ptr(exp_avg), ptr(exp_avg_sq), n, float(lr), float(betas[0]),
                           float(betas[1]), float(eps), float(weight_decay), int(step), ptr(gnorm_sq), float(max_grad_norm), stream_handle()), "adamw")


def wgrad_direct_ok(dy, x, out):
    """True if lmx_op_gemm_wgrad takes this call (16-bit, out / in features multiples of 256, rows a multiple of 64, aligned): no transposed copies needed."""
    if dy.dtype == torch.float32 or dy.dtype != x.dtype or out.dtype != dy.dtype or dy.stride(1) != 1 or x.stride(1) != 1 or out.stride(1) != 1:
        return False
    if (dy.data_ptr() | x.data_ptr() | out.data_ptr()) & 15:
        return False
    return bool(lib.lmx_op_gemm_wgrad_supported(torch_dtype_code(dy.dtype), dy.stride(0), x.stride(0), dy.shape[0], dy.shape[1], x.shape[1], out.stride(0)))


def gemm_wgrad(dy, x, out):
    """out [O, I] = dy[R, O]^T @ x[R, I] with both operands in their forward layout (csrc/gemm8t.hip); fails loudly on shapes wgrad_direct_ok rejects."""
    _need_cuda_rows(dy, x, out)
    assert dy.shape[0] == x.shape[0] and tuple(out.shape) == (dy.shape[1], x.shape[1])
    check(lib.lmx_op_gemm_wgrad(torch_dtype_code(dy.dtype), ptr(dy), dy.stride(0), ptr(x), x.stride(0), dy.shape[0], dy.shape[1], x.shape[1], ptr(out), out.stride(0),
                                stream_handle()), "gemm_wgrad")
    return out


def transpose_padded(x, multiple):
    """x [r, c] -> [c, roundup(r, multiple)] with zero columns past r: the operand layout of a GEMM that contracts over r."""
    _need_cuda_rows(x)
    r, c = x.shape
    rp = -(-r // multiple) * multiple
    out = torch.empty((c, rp), dtype=x.dtype, device=x.device) if rp == r else torch.zeros((c, rp), dtype=x.dtype, device=x.device)
    check(lib.lmx_op_transpose(torch_dtype_code(x.dtype), ptr(x), x.stride(0), r, c, ptr(out), out.stride(0), stream_handle()), "transpose")
    return out
