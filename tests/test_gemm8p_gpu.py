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
@pytest.mark.parametrize("variant", [30, 31, 32, 33, 34, 35])
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
    for variant in (30, 32, 33):
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
    """The in-launch reducer adds the partial tiles in slice order whoever arrives last: bit-identical repeats, and equal to the
    unsplit kernel wherever fp32 association does not move the sum across a bf16 rounding boundary."""
    from llava_mi355x import ops
    x, w = _mk(1087, 4096, 4096, "bf16", cuda, 11)
    outs = [ops.gemm(x, w, variant=34) for _ in range(6)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    one = ops.gemm(x, w, variant=35)
    assert (one != outs[0]).float().mean().item() < 5e-3


def test_gemm8p_split_under_uneven_load(cuda):
    """The partial-tile hand-off between the K slices of a tile under UNEVEN load (an unrelated GEMM on a second stream shares the chip)
    with the scratch reused launch after launch: the conditions under which a publish protocol that looks fine on an idle chip returns
    stale partial tiles (measured: plain stores + release / acquire + plain loads failed 44 of 96 launches here; guide §6 G16
    "test every hand-off under uneven load")."""
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


@pytest.mark.parametrize("M,N,K,act", [(600, 4096, 2048, 0), (512, 4096, 2048, 0), (1087, 5120, 2048, 0), (700, 4096, 4096, 3)])
def test_gemm8p_tail_split_order(cuda, monkeypatch, M, N, K, act):
    """Tail-split order of the ping-pong kernel (variant 36 / automatic when a launch has more full tiles than CUs): per XCD whole tiles first, the last
    full tiles as two K-halves with the in-launch reduction, ragged M-tiles last.  LMX_GEMM8P_TAIL_CUS pretends an XCD has 2 CUs so that shapes this
    small take the path: results within one ulp of the fp64 product, bit-identical repeats, and every element equal to the plain order's wherever fp32
    association does not cross a rounding boundary; with a ragged last M-tile, without one, with the SiLU*mul epilogue after the reduction."""
    from llava_mi355x import _C, ops
    monkeypatch.setenv("LMX_GEMM8P_TAIL_CUS", "2")
    x, w = _mk(M, N, K, "bf16", cuda, M + N)
    if act == 3:
        I = N // 2
        g_ = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16(); u_ = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16()
        w = ops.interleave_gate_up(g_, u_)
        ref = torch.nn.functional.silu(x.double() @ g_.double().t()) * (x.double() @ u_.double().t())
        kw = dict(act=_C.ACT_SILU_MUL)
    else:
        ref = x.double() @ w.double().t()
        kw = {}
    tail = [ops.gemm(x, w, variant=36, **kw) for _ in range(4)]
    plain = ops.gemm(x, w, variant=35, **kw)
    assert_one_ulp(tail[0], ref, "bf16", 3e-5 * float(ref.abs().max()), f"tail split {M}x{N}x{K}")
    for t in tail[1:]:
        assert torch.equal(t, tail[0])
    assert (tail[0] != plain).float().mean().item() < 5e-3
    monkeypatch.delenv("LMX_GEMM8P_TAIL_CUS")
    assert torch.equal(ops.gemm(x, w, variant=36, **kw), plain)            # at the real CU count these shapes do not qualify: the plain order runs


@pytest.mark.parametrize("M,N,K,act", [(600, 4096, 2048, 0), (512, 4096, 2048, 0), (1087, 5120, 2048, 0), (700, 4096, 4096, 3), (1087, 2312, 192, 0)])
def test_gemm8p_m_tail_order(cuda, monkeypatch, M, N, K, act):
    """M-tail order of the un-split ping-pong kernel (variant 37 / automatic under LMX_GEMM8P_MTAIL=1 when the last round of full tiles would fill at most half
    of the CUs): per XCD whole tiles first, the remaining full tiles as two 128-row halves (both wave groups keep working: 64 rows each), ragged M-tiles
    last.  No partial sums leave a workgroup and every element is accumulated over K in the same order, so the result must be BIT-IDENTICAL to the plain
    order (variant 35).  LMX_GEMM8P_TAIL_CUS pretends an XCD has 2 CUs so that small shapes take the path; with and without a ragged last M-tile, with a
    residual, with the SiLU*mul epilogue, with N not a multiple of 256."""
    from llava_mi355x import _C, ops
    monkeypatch.setenv("LMX_GEMM8P_TAIL_CUS", "2")
    x, w = _mk(M, N, K, "bf16", cuda, 3 * M + N)
    if act == 3:
        I = N // 2
        g_ = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16(); u_ = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16()
        w = ops.interleave_gate_up(g_, u_)
        ref = torch.nn.functional.silu(x.double() @ g_.double().t()) * (x.double() @ u_.double().t())
        kw = dict(act=_C.ACT_SILU_MUL)
    else:
        ref = x.double() @ w.double().t()
        kw = {}
    plain = ops.gemm(x, w, variant=35, **kw)
    got = ops.gemm(x, w, variant=37, **kw)
    assert_one_ulp(got, ref, "bf16", 3e-5 * float(ref.abs().max()), f"M-tail {M}x{N}x{K}")
    assert torch.equal(got, plain)
    if act == 0:
        r = torch.randn(M, N, device=cuda).bfloat16()
        assert torch.equal(ops.gemm(x, w, residual=r, variant=37), ops.gemm(x, w, residual=r, variant=35))
    # rows / columns outside the problem are never written: the output buffer keeps its sentinel there
    if act == 0:
        big = torch.full((M + 8, N + 8), 7.0, device=cuda, dtype=torch.bfloat16)
        ops.gemm(x, w, variant=37, out=big[:M, :N])
        assert torch.equal(big[:M, :N], plain) and bool((big[M:] == 7).all()) and bool((big[:, N:] == 7).all())


def test_gemm8p_m_tail_real_gate_up_shape(cuda, monkeypatch):
    """7B gate|up of the config-2 prefill (1087 x 22016 x 4096, SiLU*mul epilogue: 344 full + 86 ragged tiles on 256 CUs) at the real CU count: the M-tail
    order is what the automatic rule picks (LMX_GEMM8P_MTAIL=1), bit-identical to the plain order."""
    from llava_mi355x import _C, ops
    M, I, K = 1087, 11008, 4096
    g = torch.Generator(device=cuda); g.manual_seed(5)
    x = torch.randn(M, K, device=cuda, generator=g).bfloat16()
    g_ = (torch.randn(I, K, device=cuda, generator=g) / math.sqrt(K)).bfloat16(); u_ = (torch.randn(I, K, device=cuda, generator=g) / math.sqrt(K)).bfloat16()
    w = ops.interleave_gate_up(g_, u_)
    plain = ops.gemm(x, w, variant=35, act=_C.ACT_SILU_MUL)
    forced = ops.gemm(x, w, variant=37, act=_C.ACT_SILU_MUL)
    assert torch.equal(forced, plain)
    ref = torch.nn.functional.silu(x[:64].double() @ g_.double().t()) * (x[:64].double() @ u_.double().t())
    assert_one_ulp(plain[:64], ref, "bf16", 3e-5 * float(ref.abs().max()), "gate|up rows 0..63")


def test_gemm8p_tail_split_under_uneven_load(cuda, monkeypatch):
    from llava_mi355x import ops
    monkeypatch.setenv("LMX_GEMM8P_TAIL_CUS", "2")
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=cuda)
    bad = []
    for it in range(24):
        x, w = _mk(1087, 4096, 2048, "bf16", cuda, 77 * it)
        ref = ops.gemm(x, w, variant=35)
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk @ junk
        got = ops.gemm(x, w, variant=36)
        torch.cuda.synchronize()
        d = (got.float() - ref.float()).abs().max().item()
        if d > 2.0 ** -6 * ref.float().abs().max().item():
            bad.append((it, d))
    assert not bad, bad


def test_fused_rmsnorm_in_the_splitk_reduction(cuda):
    """o_proj / down_proj of a ~1k-row prefill run K-sliced with the launch-boundary reduction, which also writes LlamaRMSNorm of the rows it produces
    (HF5:models/llama/modeling_llama.py:53-67, 284-325): the next block's input.  The residual stream it writes is the same value as the unfused
    sequence's (LMX_FUSE_NORM=0: reduction + rmsnorm launch); its sum of squares is reduced in another (fixed) order, so the normalised rows may differ
    in the last bf16 bit of a few elements.  Two opt-in launch forms, both measured slower than the separate launch (LMX_FUSE_NORM=2: tile-shaped reduction + in-launch exchange of the row sums; =1: row-owning
    reduction).  LLaVA-1.5-7B widths, 1087 positions, 3 layers: the fused launch replaces 5 of the 6 rmsnorm launches, logits
    agree to bf16 noise (far inside the engine-vs-oracle tolerance of tests/test_full_depth_gpu.py) and repeat bit-identically, greedy ids agree."""
    import os
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 3, 1)
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, max_position=2048)
    ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(cuda, torch.bfloat16)
    outs = {}
    old = os.environ.get("LMX_FUSE_NORM")
    try:
        for mode in ("2", "0", "2b", "1", "3", "3b"):
            os.environ["LMX_FUSE_NORM"] = mode[0]
            model.profile(True)
            o = model.forward(input_ids=ids, images=pix, use_cache=False)
            names = model.profile_read()
            model.profile(False)
            g = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=6, eos_token_id=-1)
            outs[mode] = (o.logits.float().cpu(), g.cpu(), names.get("prefill.rmsnorm", (0.0, 0))[1])
    finally:
        if old is None:
            os.environ.pop("LMX_FUSE_NORM", None)
        else:
            os.environ["LMX_FUSE_NORM"] = old
    # fused: only layer 0's first norm is a launch of its own (3 layers -> 1 launch instead of 6).  Mode 2 (round 4) = the tile-shaped reduction
    # whose N-tiles exchange their partial sums of squares inside the launch; mode 1 = round 3's row-owning reduction
    assert outs["2"][2] == 1 and outs["1"][2] == 1 and outs["3"][2] == 1 and outs["0"][2] == 6, (outs["2"][2], outs["1"][2], outs["3"][2], outs["0"][2])
    assert torch.equal(outs["2"][0], outs["2b"][0])                      # deterministic: partials are summed in tile order whoever arrives last
    assert torch.equal(outs["3"][0], outs["3b"][0])
    scale = outs["0"][0].abs().max().item()
    assert not torch.isnan(outs["2"][0]).any()                           # a timed-out exchange would poison the rows
    for m in ("2", "1", "3"):
        err = (outs[m][0] - outs["0"][0]).abs().max().item()
        assert err <= 8e-3 * scale, (m, err, scale)                      # a few last-bit flips of bf16 activations, 3 layers deep
        assert torch.equal(outs[m][1], outs["0"][1])
