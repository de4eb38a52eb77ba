"""lmx_op_gemm_wgrad (csrc/gemm8t.hip): the weight half of nn.Linear's backward, grad_weight = grad_output^T @ input (torch autograd under the reference's
training step, llava/train/train.py:780-1000), computed from dy [rows][out] and x [rows][in] in their FORWARD layout — the contraction index is the slow
index of both operands and the MFMA fragments come out of LDS through ds_read_b64_tr_b16.

Checked here, through the C ABI:
  * against float64 (16-bit inputs are exact in fp64; one rounding of an fp32 sum to the 16-bit output),
  * BIT-IDENTICAL to the path it replaces at the training shapes (transpose both operands, the un-split ping-pong kernel of lmx_op_gemm): same MFMA, same values
    in the same contraction slots, same K-step order (small problems, where lmx_op_gemm slices K or takes its 64 x 64 kernel, sum in another order: the whole-step
    test below holds those to a rounding-level tolerance),
  * operands that are column windows of wider buffers (the step passes dq|dk|dv and its slices), one and many K-steps, odd / even K-step counts (the ring's two
    parities and its tail), more tiles than CUs,
  * inputs whose transpose differs from them everywhere (a transposed read of either operand cannot pass),
  * the shapes it does not take are refused loudly and `wgrad_direct_ok` says so beforehand,
  * a whole TrainStep with and without it: every gradient within a 16-bit rounding of the other path's."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from llava_mi355x import ops
    return ops


def _rand(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dtype).to(dev)


SHAPES = [
    # rows, out features, in features
    (64, 256, 256),          # one K-step, one tile
    (128, 256, 512),         # two K-steps
    (192, 512, 256),         # three K-steps: the odd tail of the ring
    (320, 256, 256),         # five
    (1024, 768, 256),        # q|k|v of the tiny geometry
    (2048, 1024, 768),       # 12 tiles
    (512, 4096, 4352),       # 272 tiles: more workgroups than CUs, the XCD remap's ragged end
    (4096, 12288, 4096),     # q|k|v of the 7B geometry, two samples' rows
    (2048, 4096, 11008),     # down_proj of the 7B geometry, one sample's rows
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,O,I", SHAPES)
def test_wgrad_matches_fp64_and_the_two_transpose_path(cuda, rows, O, I, dtype):
    ops = _ops()
    dy = _rand((rows, O), dtype, cuda, 1, 0.5)
    x = _rand((rows, I), dtype, cuda, 2, 0.5)
    out = torch.full((O, I), float("nan"), dtype=dtype, device=cuda)
    assert ops.wgrad_direct_ok(dy, x, out)
    ops.gemm_wgrad(dy, x, out)
    ref = dy.double().t() @ x.double()
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * ref.abs().max().item() + 1e-6       # one rounding of the output + the fp32 sum's slack
    assert torch.isfinite(out.float()).all()
    assert (out.double() - ref).abs().max().item() <= tol
    old = ops.gemm(ops.transpose_padded(dy, 64), ops.transpose_padded(x, 64), variant=35)     # 35: the un-split ping-pong kernel whatever the tile count
    assert torch.equal(out, old)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wgrad_on_column_windows_of_wider_buffers(cuda, dtype):
    """dy = the k | v columns of a [rows][q|k|v] buffer, x = a window of a wider activation buffer, out = rows of a wider gradient buffer."""
    ops = _ops()
    rows, O, I = 384, 512, 256
    big_dy = _rand((rows, 1280), dtype, cuda, 3); big_x = _rand((rows, 1024), dtype, cuda, 4)
    dy, x = big_dy[:, 768:768 + O], big_x[:, 512:512 + I]
    big_out = torch.zeros((O, 640), dtype=dtype, device=cuda)
    out = big_out[:, 128:128 + I]
    assert ops.wgrad_direct_ok(dy, x, out)
    ops.gemm_wgrad(dy, x, out)
    ref = dy.double().t() @ x.double()
    assert (out.double() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item()
    assert torch.equal(out, ops.gemm(ops.transpose_padded(dy.contiguous(), 64), ops.transpose_padded(x.contiguous(), 64), variant=35))
    assert big_out[:, :128].abs().max().item() == 0 and big_out[:, 128 + I:].abs().max().item() == 0      # nothing outside the window was written


def test_wgrad_structured_operands_catch_a_transposed_read(cuda):
    """dy[r][o] = small integer code of (r, o), x one-hot in r: out[o][i] = dy[r_i][o] exactly — any mix-up of row / column / contraction slot shows as a wrong integer."""
    ops = _ops()
    rows, O, I = 256, 256, 256
    r = torch.arange(rows, device=cuda)[:, None]; o = torch.arange(O, device=cuda)[None, :]
    dy = ((r * 3 + o * 5) % 251).to(torch.bfloat16)                     # integers < 256: exact in bf16
    perm = torch.from_numpy(np.random.RandomState(0).permutation(rows)).to(cuda)
    x = torch.zeros((rows, I), dtype=torch.bfloat16, device=cuda)
    x[perm, torch.arange(I, device=cuda)] = 1.0                         # column i picks row perm[i]
    out = torch.empty((O, I), dtype=torch.bfloat16, device=cuda)
    ops.gemm_wgrad(dy, x, out)
    want = dy[perm, :].t().contiguous()                                 # out[o][i] = dy[perm[i]][o]
    assert torch.equal(out, want)


def test_wgrad_refuses_what_it_does_not_take(cuda):
    ops = _ops()
    from llava_mi355x._C import LmxError
    ok = lambda r, O, I, dt=torch.bfloat16: ops.wgrad_direct_ok(torch.empty((r, O), dtype=dt, device=cuda), torch.empty((r, I), dtype=dt, device=cuda),
                                                                torch.empty((O, I), dtype=dt, device=cuda))
    assert ok(64, 256, 256)
    assert not ok(64, 128, 256) and not ok(64, 256, 320) and not ok(32, 256, 256) and not ok(64, 256, 256, torch.float32)
    dy = torch.zeros((64, 128), dtype=torch.bfloat16, device=cuda); x = torch.zeros((64, 256), dtype=torch.bfloat16, device=cuda)
    with pytest.raises(LmxError):
        ops.gemm_wgrad(dy, x, torch.empty((128, 256), dtype=torch.bfloat16, device=cuda))


def test_train_step_gradients_agree_with_and_without_it(cuda):
    from synthetic import recipes as synth
    from test_train_step_gpu import build_step, make_batch, reference_step
    cfg = synth.CONFIGS["tiny"]                                          # hidden 256, intermediate 512, q|k|v 768, vocabulary 512: every decoder linear qualifies
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg)
    ref = reference_step(cfg, wnp, batch, 1e-3, 0.0, 1.0)
    ids, mask, labels, _ = batch
    a = build_step(cfg, wnp, torch.bfloat16, cuda, direct_wgrad=True)
    b = build_step(cfg, wnp, torch.bfloat16, cuda, direct_wgrad=False)
    calls = {"n": 0}
    from llava_mi355x import ops
    real = ops.gemm_wgrad
    def counted(*args, **kw):
        calls["n"] += 1
        return real(*args, **kw)
    ops.gemm_wgrad = counted
    try:
        la, _ = a.forward_backward(ids, labels, mask, image_features=ref["tower"])
        n_direct = calls["n"]
        lb, _ = b.forward_backward(ids, labels, mask, image_features=ref["tower"])
    finally:
        ops.gemm_wgrad = real
    assert n_direct >= 5 * a.L + 1 and calls["n"] == n_direct           # q|k|v, o, gate, up, down per layer + lm_head took it; the second step never did
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(lb.item())           # the forward is the same code
    for k in a.g:                                                        # tiny problems: the replaced path slices K / takes the 64 x 64 kernel -> another summation order
        ga, gb = a.g[k].float(), b.g[k].float()
        assert (ga - gb).abs().max().item() <= 2.0 ** -7 * gb.abs().max().item() + 1e-12, k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(64, 128), (8, 8), (72, 136), (200, 328), (2048, 4096), (1087, 768), (33, 77), (4096, 11008)])
def test_transpose_padded_is_exact(cuda, rows, cols, dtype):
    """lmx_op_transpose (csrc/train.hip: the swizzled 64 x 128 tile for 16-bit operands with every extent a multiple of 8, the scalar 64 x 64 tile otherwise): every
    element lands, ragged tile edges included, the padding columns of the output are zero, and a source that is a column window of a wider buffer reads its own columns."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(rows * 131 + cols)
    wide = torch.randn((rows, cols + 24), generator=g, dtype=torch.float32).to(dtype).to(cuda)
    for src in (wide[:, :cols].contiguous(), wide[:, 16:16 + cols]):
        out = ops.transpose_padded(src, 64)
        rp = -(-rows // 64) * 64
        assert tuple(out.shape) == (cols, rp)
        assert torch.equal(out[:, :rows], src.t())
        assert rp == rows or out[:, rows:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,H", [(1, 4096), (16, 4096), (129, 4096), (200, 2048), (640, 1024), (2048, 4096), (37, 512)])
@pytest.mark.parametrize("with_res", [False, True])
def test_rmsnorm_bwd_one_pass_form(cuda, rows, H, dtype, with_res):
    """lmx_op_rmsnorm_bwd / lmx_op_rmsnorm_bwd_add (csrc/train.hip rmsnorm_bwd_fused_kernel for 16-bit rows of 1024 / 2048 / 4096; H = 512: the three-launch form)
    against the fp64 formula of LlamaRMSNorm's autograd (HF5:models/llama/modeling_llama.py:53-67) on the 16-bit inputs:
        inv = rsqrt(mean(x^2) + eps), xhat = x inv, g = dy w, dx = inv (g - xhat mean(g xhat)), dw = sum_rows dy xhat;   with a residual: dx = T(res + T(dx)).
    Row counts that are not multiples of the 16 rows a wave walks or the 128 a workgroup owns; dx-only calls (no dw) too."""
    ops = _ops()
    eps = 1e-5
    x = _rand((rows, H), dtype, cuda, 11, 1.3); dy = _rand((rows, H), dtype, cuda, 12, 0.7)
    w = (1 + 0.1 * _rand((H,), torch.float32, cuda, 13)).to(dtype)
    res = _rand((rows, H), dtype, cuda, 14, 0.9) if with_res else None
    xd, gd, wd = x.double(), dy.double(), w.double()
    inv = torch.rsqrt((xd * xd).mean(-1, keepdim=True) + eps)
    xh = xd * inv; g = gd * wd
    dx_ref = inv * (g - xh * (g * xh).mean(-1, keepdim=True))
    dw_ref = (gd * xh).sum(0)
    dx, dw = ops.rmsnorm_bwd(x, w, dy, eps, residual=res)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    if with_res:
        want = (res.double() + dx_ref.to(dtype).double())
        assert (dx.double() - want).abs().max().item() <= 2 * ulp * want.abs().max().item()
    else:
        assert (dx.double() - dx_ref).abs().max().item() <= 1.01 * ulp * dx_ref.abs().max().item()
    assert (dw.double() - dw_ref).abs().max().item() <= 1e-5 * dw_ref.abs().max().item() + 1e-6
    dx2, none = ops.rmsnorm_bwd(x, w, dy, eps, want_dw=False, residual=res)
    assert none is None and torch.equal(dx2, dx)
    dx3, dw3 = ops.rmsnorm_bwd(x, w, dy, eps, residual=res)
    assert torch.equal(dx3, dx) and torch.equal(dw3, dw)                 # deterministic: partial rows are added in a fixed order


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nh,nkv,D", [(32, 32, 128), (4, 2, 64)])
def test_rope_kv_rows_leaves_rotated_k_in_the_rows(cuda, nh, nkv, D, dtype):
    """lmx_op_rope_kv_rows = lmx_op_rope_kv (q rotated in place, rotated k / v appended to the caches) + the rotated k written over the k columns of qkv:
    the same q columns, the same caches, v columns untouched, and the k columns equal to the cache rows bit for bit (the training backward reads k from qkv)."""
    ops = _ops()
    T = 77
    table = torch.randn((128, D), dtype=torch.float32, device=cuda)            # any cos | sin table: the two entry points must agree on it
    qkv0 = _rand((T, (nh + 2 * nkv) * D), dtype, cuda, 21)
    a, b = qkv0.clone(), qkv0.clone()
    kc_a, vt_a = ops.alloc_kv(nkv, 128, D, dtype, cuda); kc_b, vt_b = ops.alloc_kv(nkv, 128, D, dtype, cuda)
    ops.rope_kv(a, kc_a, vt_a, table, 3, nh, nkv, D)
    ops.rope_kv(b, kc_b, vt_b, table, 3, nh, nkv, D, k_rows=True)
    assert torch.equal(kc_a[:, 3:3 + T], kc_b[:, 3:3 + T]) and torch.equal(vt_a[:, :, 3:3 + T], vt_b[:, :, 3:3 + T])
    assert torch.equal(a[:, :nh * D], b[:, :nh * D]) and torch.equal(b[:, (nh + nkv) * D:], qkv0[:, (nh + nkv) * D:])
    assert torch.equal(a[:, nh * D:(nh + nkv) * D], qkv0[:, nh * D:(nh + nkv) * D])                      # the plain entry point leaves k alone
    assert torch.equal(b[:, nh * D:(nh + nkv) * D], kc_b[:, 3:3 + T].permute(1, 0, 2).reshape(T, nkv * D))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(64, 512), (37, 11008), (5, 13), (1, 8)])
def test_swiglu_bwd_16_byte_form(cuda, shape, dtype):
    """lmx_op_swiglu_bwd for 16-bit tensors moves 8 elements per lane (plus a scalar tail when the count is not a multiple of 8): autograd of silu(g) * u
    (HF5:models/llama/modeling_llama.py:163-176) in float64 on the 16-bit inputs, one rounding of the result."""
    ops = _ops()
    g = _rand(shape, dtype, cuda, 31, 2.0); u = _rand(shape, dtype, cuda, 32); d = _rand(shape, dtype, cuda, 33)
    dg, du = ops.swiglu_bwd(g, u, d)
    gd, ud, dd = g.double(), u.double(), d.double()
    sg = torch.sigmoid(gd)
    want_g = dd * ud * sg * (1 + gd * (1 - sg)); want_u = dd * gd * sg
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for got, want in ((dg, want_g), (du, want_u)):
        assert ((got.double() - want).abs() <= 1.01 * ulp * want.abs() + 1e-6).all()


def _attn_setup(cuda, T, nh, nkv, D, dtype, seed):
    """Rotated q | k | v rows of one sample in the caches and as rows; returns (qkv rows, kc, vt, q, k, v as float64 [heads][T][D])."""
    ops = _ops()
    qkv = _rand((T, (nh + 2 * nkv) * D), dtype, cuda, seed, 0.8)
    kc, vt = ops.alloc_kv(nkv, -(-T // 128) * 128, D, dtype, cuda)
    ops.rope_kv(qkv, kc, vt, torch.cat([torch.ones(T + 1, D // 2), torch.zeros(T + 1, D // 2)], -1).to(cuda), 0, nh, nkv, D, k_rows=True)   # cos 1, sin 0: identity
    q = qkv[:, :nh * D].reshape(T, nh, D).permute(1, 0, 2).double()
    k = qkv[:, nh * D:(nh + nkv) * D].reshape(T, nkv, D).permute(1, 0, 2).double()
    v = qkv[:, (nh + nkv) * D:].reshape(T, nkv, D).permute(1, 0, 2).double()
    return qkv, kc, vt, q, k, v


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,nh,nkv,D", [(200, 4, 4, 128), (129, 4, 2, 128), (64, 2, 2, 64), (1000, 8, 8, 128)])
def test_flash_attn_lse_is_the_log_sum_exp_of_the_scaled_scores(cuda, T, nh, nkv, D, dtype):
    """lmx_op_flash_attn_lse: the same output bits as lmx_op_flash_attn, and lse[h][i] = log2 sum_{j <= i} 2^(log2(e) q_i . k_j / sqrt(D)) against float64."""
    ops = _ops()
    qkv, kc, vt, q, k, v = _attn_setup(cuda, T, nh, nkv, D, dtype, 41)
    Tp = -(-T // 64) * 64
    lse = torch.zeros((nh, Tp), dtype=torch.float32, device=cuda)
    out = ops.flash_attn(qkv, kc, vt, T, T, 0, nh, nkv, D, True, lse=lse)
    plain = ops.flash_attn(qkv, kc, vt, T, T, 0, nh, nkv, D, True)
    assert torch.equal(out, plain)
    kk = k.repeat_interleave(nh // nkv, 0)
    s = (q @ kk.transpose(1, 2)) / math.sqrt(D)
    s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device=cuda)), float("-inf"))
    want = torch.logsumexp(s, -1) / math.log(2.0)
    assert (lse[:, :T].double() - want).abs().max().item() <= 2e-3                 # fp32 sums of 16-bit products
    assert lse[:, T:].abs().max().item() == 0 if Tp > T else True                  # rows past q_len are not written


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,nh,nkv,D", [(200, 4, 4, 128), (129, 4, 2, 128), (64, 2, 2, 64), (777, 8, 8, 128)])
def test_attn_bwd_with_the_forwards_statistics(cuda, T, nh, nkv, D, dtype):
    """lmx_op_attn_bwd_lse (the forward's lse and output in, no statistics sweep; delta = rowsum(dO o O)) against torch autograd in float64 on the 16-bit q, k, v, dO,
    and against lmx_op_attn_bwd (statistics recomputed in the backward) — the two differ by the rounding of O inside delta only."""
    ops = _ops()
    qkv, kc, vt, q, k, v = _attn_setup(cuda, T, nh, nkv, D, dtype, 43)
    lse = torch.full((nh, -(-T // 64) * 64), float("nan"), dtype=torch.float32, device=cuda)          # the entries past T stay NaN: the backward must not let them through
    out = ops.flash_attn(qkv, kc, vt, T, T, 0, nh, nkv, D, True, lse=lse)
    d_out = _rand((T, nh * D), dtype, cuda, 44)
    qr = qkv[:, :nh * D].contiguous(); kr = qkv[:, nh * D:(nh + nkv) * D].contiguous(); vr = qkv[:, (nh + nkv) * D:].contiguous()
    dq, dk, dv = ops.attn_bwd_lse(qr, kr, vr, out, d_out, lse, nh, nkv, D)
    dq0, dk0, dv0 = ops.attn_bwd(qr, kr, vr, d_out, nh, nkv, D)
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    g = nh // nkv
    s = (qd @ kd.repeat_interleave(g, 0).transpose(1, 2)) / math.sqrt(D)
    s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device=cuda)), float("-inf"))
    o = torch.softmax(s, -1) @ vd.repeat_interleave(g, 0)
    o.backward(d_out.double().reshape(T, nh, D).permute(1, 0, 2))
    rows = lambda t, h: t.permute(1, 0, 2).reshape(T, h * D)
    tol = 3e-2 if dtype == torch.bfloat16 else 6e-3
    for name, got, old, want in (("dq", dq, dq0, rows(qd.grad, nh)), ("dk", dk, dk0, rows(kd.grad, nkv)), ("dv", dv, dv0, rows(vd.grad, nkv))):
        scale = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= tol * scale, name
        assert (got.double() - old.double()).abs().max().item() <= tol * scale, name + " vs the recomputing form"
