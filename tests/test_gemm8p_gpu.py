"""Ping-pong prefill GEMM (csrc/gemm8p.hip, variants 30-35 of lmx_op_gemm) and the per-op 1-ulp checks at the real LLaVA-1.5-7B
prefill shapes (T = 1087): every 16-bit GEMM output must equal the fp64 product rounded ONCE to the storage dtype, up to one
unit in the last place (fp32 accumulation order can move a sum across a rounding boundary; nothing larger is accepted).

Reference arithmetic: torch.nn.Linear as reached from HF5:models/llama/modeling_llama.py:163-176,243-281 (fp32-accumulated GEMM,
one rounding to the model dtype per Linear output)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
MANT = {"bf16": 8, "f16": 11}          # significand bits incl. the hidden one


def ulp_of(ref64: torch.Tensor, dt: str) -> torch.Tensor:
    """Spacing of the storage dtype at |ref| (fp64 tensor), clamped at the smallest normal."""
    tiny = torch.finfo(DT[dt]).tiny
    e = torch.floor(torch.log2(ref64.abs().clamp_min(tiny)))
    return torch.pow(2.0, e - (MANT[dt] - 1))


def assert_one_ulp(got: torch.Tensor, ref64: torch.Tensor, dt: str, acc_abs: float, what: str = ""):
    """|got - round(ref64)| <= 1 ulp(ref) + acc_abs, where acc_abs bounds the fp32 accumulation-order noise (it only matters for
    results that cancel to ~0, whose own ulp is far below that noise)."""
    want = ref64.to(DT[dt]).double()
    err = (got.double() - want).abs()
    bound = ulp_of(ref64, dt) + acc_abs
    bad = err > bound
    n_bad = int(bad.sum().item())
    assert n_bad == 0, f"{what}: {n_bad} of {err.numel()} outputs are more than 1 ulp from the once-rounded fp64 product; worst {float((err / bound).max()):.2f}x"
    return float((err > acc_abs).double().mean().item())       # fraction that is not bit-identical-ish (informational)


def _mk(M, N, K, dt, cuda, seed):
    g = torch.Generator(device=cuda); g.manual_seed(seed)
    x = torch.randn(M, K, device=cuda, generator=g).to(DT[dt])
    w = (torch.randn(N, K, device=cuda, generator=g) / math.sqrt(K)).to(DT[dt])
    return x, w


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("variant", [30, 33, 34, 35])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 256, 128), (1087, 512, 192), (577, 384, 640), (33, 136, 320), (300, 1024, 1024),
                                   (513, 776, 2048)])
def test_gemm8p_small(cuda, dt, variant, M, N, K):
    """Edge shapes: one K-step, odd / even K-step counts (ring parity), ragged M and N tails, fewer K-steps than slices."""
    from llava_mi355x import ops
    x, w = _mk(M, N, K, dt, cuda, M * 7 + N + K)
    got = ops.gemm(x, w, variant=variant)
    ref = x.double() @ w.double().t()
    assert_one_ulp(got, ref, dt, 2e-5 * float(ref.abs().max()), f"variant {variant} {M}x{N}x{K}")


def test_gemm8p_transpose_detecting(cuda):
    from llava_mi355x import ops
    M = N = 256; K = 128
    x = torch.eye(M, K, device=cuda, dtype=torch.bfloat16)
    w = (torch.arange(N, device=cuda).float()[:, None] * 0.25 + torch.arange(K, device=cuda).float()[None, :] * 0.001953125).to(torch.bfloat16)
    for variant in (30, 33):
        got = ops.gemm(x, w, variant=variant)
        assert torch.equal(got.float(), (x.float() @ w.float().t()).to(torch.bfloat16).float())


@pytest.mark.parametrize("variant", [30, 33, 34])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm8p_bias_act_residual(cuda, variant, act):
    from llava_mi355x import ops
    M, N, K = 700, 520, 1152
    x, w = _mk(M, N, K, "bf16", cuda, act + variant)
    b = torch.randn(N, device=cuda).bfloat16(); r = torch.randn(M, N, device=cuda).bfloat16()
    y = x.double() @ w.double().t() + b.double()
    if act == 1:
        y = y * torch.sigmoid(1.702 * y)
    elif act == 2:
        y = torch.nn.functional.gelu(y)
    ref = y + r.double()
    got = ops.gemm(x, w, bias=b, residual=r, act=act, variant=variant)
    assert_one_ulp(got, ref, "bf16", 3e-5 * float(ref.abs().max()), f"variant {variant} act {act}")
    r2 = r.clone()                                  # in place: C aliases R, the way the engine keeps the residual stream
    ops.gemm(x, w, bias=b, residual=r2, act=act, variant=variant, out=r2)
    assert torch.equal(r2, got)


@pytest.mark.parametrize("M", [1, 77, 1087])
def test_gemm8p_silu_mul(cuda, M):
    from llava_mi355x import _C, ops
    I, K = 352, 256
    g_ = torch.Generator(device=cuda); g_.manual_seed(5)
    x = torch.randn(M, K, device=cuda, generator=g_).bfloat16()
    g = (torch.randn(I, K, device=cuda, generator=g_) / math.sqrt(K)).bfloat16(); u = (torch.randn(I, K, device=cuda, generator=g_) / math.sqrt(K)).bfloat16()
    fused = ops.interleave_gate_up(g, u)
    ref = torch.nn.functional.silu(x.double() @ g.double().t()) * (x.double() @ u.double().t())
    for variant in (30, 33):
        got = ops.gemm(x, fused, act=_C.ACT_SILU_MUL, variant=variant)
        assert got.shape == (M, I)
        assert_one_ulp(got, ref, "bf16", 3e-5 * float(ref.abs().max()), f"variant {variant}")


def test_gemm8p_split_is_deterministic(cuda):
    """The launch-boundary reduction adds the partial tiles in slice order: bit-identical repeats, and equal to the
    unsplit kernel wherever fp32 association does not move the sum across a bf16 rounding boundary."""
    from llava_mi355x import ops
    x, w = _mk(1087, 4096, 4096, "bf16", cuda, 11)
    outs = [ops.gemm(x, w, variant=34) for _ in range(6)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    one = ops.gemm(x, w, variant=35)
    assert (one != outs[0]).float().mean().item() < 5e-3


def test_gemm8p_split_under_uneven_load(cuda):
    """The partial tiles of a K-sliced launch under UNEVEN load (an unrelated GEMM on a second stream shares the chip) with the scratch reused
    launch after launch: the conditions under which round 2's in-launch publish protocols returned stale partial tiles (44 of 96 launches with
    plain stores + release / acquire + plain loads).  The launch-boundary reduction of round 3 on has no hand-off inside a launch; the test stays
    as the guard of that property."""
    from llava_mi355x import ops
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=cuda)
    bad = []
    for M, N, K in ((1087, 4096, 11008), (1087, 4096, 4096), (513, 776, 2048)):
        for it in range(16):
            x, w = _mk(M, N, K, "bf16", cuda, 1000 * it + N)
            ref = ops.gemm(x, w, variant=35)
            if it % 3 == 0:
                with torch.cuda.stream(side):
                    junk @ junk
            got = ops.gemm(x, w, variant=34 if it % 2 else 33)
            torch.cuda.synchronize()
            d = (got.float() - ref.float()).abs().max().item()
            if d > 2.0 ** -6 * ref.float().abs().max().item():
                bad.append((M, N, K, it, d))
    assert not bad, bad


REAL = [("qkv", 1087, 12288, 4096), ("o_proj", 1087, 4096, 4096), ("gate_up", 1087, 22016, 4096), ("down", 1087, 4096, 11008),
        ("qkv_13b", 1087, 15360, 5120), ("down_13b", 1087, 5120, 13824)]


@pytest.mark.parametrize("name,M,N,K", REAL)
def test_real_shape_gemm_one_ulp(cuda, name, M, N, K):
    """the bf16 MFMA GEMMs the bench times (auto tile selection = variant 0, and the ping-pong kernel forced) at the
    real prefill shapes, against the fp64 product rounded once."""
    from llava_mi355x import _C, ops
    x, w = _mk(M, N, K, "bf16", cuda, N + K)
    ref = x.double() @ w.double().t()
    tol = 2e-5 * float(ref.abs().max())
    for variant in (0, 30):
        got = ops.gemm(x, w, variant=variant)
        frac = assert_one_ulp(got, ref, "bf16", tol, f"{name} variant {variant}")
        assert frac < 0.02, f"{name} variant {variant}: {frac:.4f} of the outputs differ from the once-rounded product"
    if name == "gate_up":
        g, u = w[: N // 2].contiguous(), w[N // 2:].contiguous()
        fused = ops.interleave_gate_up(g, u)
        ref2 = torch.nn.functional.silu(x.double() @ g.double().t()) * (x.double() @ u.double().t())
        got = ops.gemm(x, fused, act=_C.ACT_SILU_MUL, variant=0)
        assert_one_ulp(got, ref2, "bf16", 3e-5 * float(ref2.abs().max()), "gate|up SiLU·mul")
    if name in ("o_proj", "down"):
        r = torch.randn(M, N, device=cuda).bfloat16()
        got = ops.gemm(x, w, residual=r, variant=0)
        assert_one_ulp(got, ref + r.double(), "bf16", tol, f"{name} + residual")
