"""Training-step slices (csrc/train.hip; SURVEY §8 f-3, BASELINE config 5) against torch autograd of the oracle's own functions.

The finetuning step of the reference is HF Trainer over LlavaLlamaForCausalLM.forward(labels=...) (llava/train/train.py:805-1000,
llava_llama.py:56-99): shifted cross-entropy with IGNORE_INDEX labels (llava_arch.py:181,200), LLaMA decoder layers with causal flash
attention (llava/train/llama_flash_attn_monkey_patch.py:68-91).  Each backward kernel is checked alone, then composed into the backward
of a whole decoder layer and compared with autograd of oracle/llava_oracle.py: decoder_layer — on a tiny layer in fp32 and on one
real-width (LLaVA-1.5-7B) layer in bf16.  Tolerances: fp32 <= 2e-4 of max|ref| (atomics reorder the dK / dV sums), bf16 <= 3e-2."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(got, ref):
    return ((got.detach().float().cpu() - ref.detach().float().cpu()).abs().max() / ref.detach().float().abs().max().clamp_min(1e-9)).item()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ce_loss_matches_torch(cuda, dt):
    from llava_mi355x import ops
    torch.manual_seed(0)
    B, T, V = 3, 17, 1000
    logits = (torch.randn(B, T, V) * 3).to(dt)
    labels = torch.randint(0, V, (B, T))
    labels[0, :5] = -100; labels[1, 9:] = -100; labels[2, ::3] = -100
    lf = logits.float().requires_grad_(True)
    ref = F.cross_entropy(lf[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    ref.backward()
    loss, count, d = ops.ce_loss(logits.to(cuda), labels.to(cuda), want_grad=True)
    assert int(count.item()) == int((labels[:, 1:] != -100).sum())
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    tol = 1e-6 if dt == torch.float32 else 8e-3           # dlogits is stored in the logits dtype
    assert (d.float().cpu() - lf.grad).abs().max().item() <= tol * lf.grad.abs().max().item() + 1e-9
    assert float(d[:, -1].abs().max()) == 0.0                    # the last position has no label to predict
    # every label ignored: mean over nothing = NaN, as torch
    none = torch.full_like(labels, -100)
    loss2, count2, _ = ops.ce_loss(logits.to(cuda), none.to(cuda))
    assert math.isnan(loss2.item()) and count2.item() == 0


def test_ce_loss_real_vocab(cuda):
    from llava_mi355x import ops
    torch.manual_seed(1)
    B, T, V = 1, 40, 32000
    logits = (torch.randn(B, T, V) * 2).bfloat16().to(cuda)
    labels = torch.randint(0, V, (B, T), device=cuda); labels[0, :20] = -100
    ref = F.cross_entropy(logits.float()[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1), ignore_index=-100)
    loss, count, _ = ops.ce_loss(logits, labels)
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()) and count.item() == 20


def test_model_forward_loss_uses_the_kernel(cuda):
    """LlavaLlamaForCausalLM.forward(labels=...) returns the reference's loss (llava_llama.py:56-99), now from lmx_op_ce_loss."""
    from oracle import llava_oracle as O
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    model = harness.build_model(cfg, dtype=torch.float32, weights=wnp)
    ids = torch.from_numpy(synth.make_prompt(cfg, 14, image_positions=(5,)))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1))
    labels = ids.clone(); labels[0, :7] = -100
    with torch.no_grad():
        ref_logits, _, _, new_labels = O.llava_forward(O.to_torch_weights(wnp), cfg, ids, pix, labels=labels)
    ref = F.cross_entropy(ref_logits[:, :-1].reshape(-1, cfg.vocab_size), new_labels[:, 1:].reshape(-1), ignore_index=-100)
    out = model.forward(input_ids=ids.to(cuda), images=pix.to(cuda), labels=labels.to(cuda), use_cache=False)
    assert abs(out.loss.item() - ref.item()) <= 1e-4 * abs(ref.item())


def test_rmsnorm_swiglu_rope_bwd(cuda):
    from llava_mi355x import ops
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    torch.manual_seed(2)
    rows, H = 37, 512
    x = torch.randn(rows, H, requires_grad=True); w = (1 + 0.1 * torch.randn(H)).requires_grad_(True); dy = torch.randn(rows, H)
    y = O.rms_norm(x, w, 1e-5); y.backward(dy)
    dx, dw = ops.rmsnorm_bwd(x.detach().to(cuda), w.detach().to(cuda), dy.to(cuda), 1e-5)
    assert _rel(dx, x.grad) <= 1e-5 and _rel(dw, w.grad) <= 1e-5
    g = torch.randn(rows, H, requires_grad=True); u = torch.randn(rows, H, requires_grad=True); da = torch.randn(rows, H)
    (F.silu(g) * u).backward(da)
    dg, du = ops.swiglu_bwd(g.detach().to(cuda), u.detach().to(cuda), da.to(cuda))
    assert _rel(dg, g.grad) <= 1e-5 and _rel(du, u.grad) <= 1e-5
    cfg = synth.CONFIGS["tiny"]
    D, nh, T, pos0 = cfg.head_dim, 4, 11, 3
    q = torch.randn(1, nh, T, D, requires_grad=True)
    cos, sin = O.rope_cos_sin(cfg, torch.arange(pos0, pos0 + T)[None], torch.float32)
    qr = q * cos[:, None] + O.rotate_half(q) * sin[:, None]
    dqr = torch.randn_like(qr); qr.backward(dqr)
    table = torch.from_numpy(O.rope_table(cfg, 64)).to(cuda)
    got = ops.rope_bwd(dqr[0].transpose(0, 1).reshape(T, nh * D).contiguous().to(cuda), table, pos0, nh, D)
    assert _rel(got, q.grad[0].transpose(0, 1).reshape(T, nh * D)) <= 1e-5


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)])
def test_linear_bwd(cuda, dt, tol):
    from llava_mi355x import ops
    torch.manual_seed(3)
    M, N, K = 192, 192, 320          # the wgrad contraction runs over M: a multiple of the GEMM k-slab (training batches are padded to it)
    x = torch.randn(M, K).to(dt); w = (torch.randn(N, K) / math.sqrt(K)).to(dt); dy = torch.randn(M, N).to(dt)
    xr = x.float().requires_grad_(True); wr = w.float().requires_grad_(True)
    (xr @ wr.t()).backward(dy.float())
    dx, dw = ops.linear_bwd(x.to(cuda), w.to(cuda), dy.to(cuda))
    assert _rel(dx, xr.grad) <= tol and _rel(dw, wr.grad) <= tol
    t = ops.transpose(x.to(cuda))
    assert torch.equal(t.cpu(), x.t().contiguous())


def _attn_ref(q, k, v, nh, nkv, D):
    T = q.shape[0]
    qh = q.view(T, nh, D).transpose(0, 1); kh = k.view(T, nkv, D).transpose(0, 1).repeat_interleave(nh // nkv, 0)
    vh = v.view(T, nkv, D).transpose(0, 1).repeat_interleave(nh // nkv, 0)
    s = qh @ kh.transpose(1, 2) / math.sqrt(D)
    s = s.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(T, nh * D)


@pytest.mark.parametrize("nh,nkv,D,T", [(4, 4, 64, 33), (8, 2, 128, 70), (32, 32, 128, 2048)])      # the last: LLaVA-1.5-7B's heads at the reference's training length
def test_attn_bwd(cuda, nh, nkv, D, T):
    from llava_mi355x import ops
    torch.manual_seed(4)
    q = torch.randn(T, nh * D, requires_grad=True); k = torch.randn(T, nkv * D, requires_grad=True); v = torch.randn(T, nkv * D, requires_grad=True)
    do = torch.randn(T, nh * D)
    _attn_ref(q, k, v, nh, nkv, D).backward(do)
    dq, dk, dv = ops.attn_bwd(q.detach().to(cuda), k.detach().to(cuda), v.detach().to(cuda), do.to(cuda), nh, nkv, D)
    assert _rel(dq, q.grad) <= 2e-4 and _rel(dk, k.grad) <= 2e-4 and _rel(dv, v.grad) <= 2e-4


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nh,nkv,D,T", [(4, 4, 64, 33), (8, 2, 128, 70), (4, 4, 128, 200), (8, 2, 64, 321), (32, 32, 128, 2048)])
def test_attn_bwd_mfma(cuda, nh, nkv, D, T, dt):
    """The matrix-core backward (csrc/attn_bwd.hip; what 16-bit training runs) against autograd of the fp32 attention on the SAME 16-bit-representable
    inputs: lengths that are no multiple of the 64-row tiles / 128-row blocks, GQA groups, head_dim 64 and 128, the reference's training length.
    P and dS enter the MFMAs rounded to the model dtype (as in FlashAttention-2, the reference's backward: llama_flash_attn_monkey_patch.py:68-91), hence
    the 16-bit tolerance; the VALU kernels (fp32 models) keep 2e-4 above."""
    if dt == torch.float16 and T > 400:
        pytest.skip("one dtype at the long length")
    from llava_mi355x import ops
    torch.manual_seed(4)
    rnd = lambda *s: torch.randn(*s).to(dt).float()
    q = rnd(T, nh * D).requires_grad_(True); k = rnd(T, nkv * D).requires_grad_(True); v = rnd(T, nkv * D).requires_grad_(True)
    do = rnd(T, nh * D)
    _attn_ref(q, k, v, nh, nkv, D).backward(do)
    f = lambda t: t.detach().to(dt).to(cuda)
    dq, dk, dv = ops.attn_bwd(f(q), f(k), f(v), f(do), nh, nkv, D)
    tol = 1.5e-2 if dt == torch.bfloat16 else 3e-3
    errs = (_rel(dq, q.grad), _rel(dk, k.grad), _rel(dv, v.grad))
    assert max(errs) <= tol, errs


def _layer_backward(ops, cfg, w, i, h, dout, dt, cuda, table):
    """Backward of one decoder layer from the lmx slices only.  Forward intermediates are recomputed with torch on the host (the forward
    kernels have their own parity tests); every gradient below comes from a HIP kernel.  Returns (dh, {weight name: grad})."""
    from oracle import llava_oracle as O
    p = f"model.layers.{i}."
    T, H = h.shape[1], h.shape[2]
    nh, nkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    f = lambda t: t.to(dt).to(cuda).contiguous()
    with torch.no_grad():
        cos, sin = O.rope_cos_sin(cfg, torch.arange(T)[None], torch.float32)
        x1 = O.rms_norm(h, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)[0]
        q = F.linear(x1, w[p + "self_attn.q_proj.weight"]); k = F.linear(x1, w[p + "self_attn.k_proj.weight"]); v = F.linear(x1, w[p + "self_attn.v_proj.weight"])
        rot = lambda t, n: (t.view(T, n, D).transpose(0, 1) * cos + O.rotate_half(t.view(T, n, D).transpose(0, 1)) * sin).transpose(0, 1).reshape(T, n * D)
        qr, kr = rot(q, nh), rot(k, nkv)
        attn = _attn_ref(qr, kr, v, nh, nkv, D)
        h2 = h[0] + F.linear(attn, w[p + "self_attn.o_proj.weight"])
        x2 = O.rms_norm(h2[None], w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)[0]
        g = F.linear(x2, w[p + "mlp.gate_proj.weight"]); u = F.linear(x2, w[p + "mlp.up_proj.weight"])
        act = F.silu(g) * u
    grads = {}
    d_out = f(dout[0])
    # MLP block: out = h2 + down(act)
    d_act, grads[p + "mlp.down_proj.weight"] = ops.linear_bwd(f(act), f(w[p + "mlp.down_proj.weight"]), d_out)
    dg, du = ops.swiglu_bwd(f(g), f(u), d_act)
    dx2_g, grads[p + "mlp.gate_proj.weight"] = ops.linear_bwd(f(x2), f(w[p + "mlp.gate_proj.weight"]), dg)
    dx2_u, grads[p + "mlp.up_proj.weight"] = ops.linear_bwd(f(x2), f(w[p + "mlp.up_proj.weight"]), du)
    dx2 = (dx2_g.float() + dx2_u.float()).to(dt)
    dh2_n, grads[p + "post_attention_layernorm.weight"] = ops.rmsnorm_bwd(f(h2), f(w[p + "post_attention_layernorm.weight"]), dx2, cfg.rms_norm_eps)
    dh2 = (d_out.float() + dh2_n.float()).to(dt)
    # attention block: h2 = h + o_proj(attn)
    d_attn, grads[p + "self_attn.o_proj.weight"] = ops.linear_bwd(f(attn), f(w[p + "self_attn.o_proj.weight"]), dh2)
    dqr, dkr, dv = ops.attn_bwd(f(qr), f(kr), f(v), d_attn, nh, nkv, D)
    dq = ops.rope_bwd(dqr, table, 0, nh, D); dk = ops.rope_bwd(dkr, table, 0, nkv, D)
    dx1 = torch.zeros((T, H), dtype=torch.float32, device=cuda)
    for nm, dgrad in (("q", dq), ("k", dk), ("v", dv)):
        dxp, grads[p + f"self_attn.{nm}_proj.weight"] = ops.linear_bwd(f(x1), f(w[p + f"self_attn.{nm}_proj.weight"]), dgrad)
        dx1 += dxp.float()
    dh_n, grads[p + "input_layernorm.weight"] = ops.rmsnorm_bwd(f(h[0]), f(w[p + "input_layernorm.weight"]), dx1.to(dt), cfg.rms_norm_eps)
    return (dh2.float() + dh_n.float()), grads


@pytest.mark.parametrize("name,dt,T,tol", [("tiny", torch.float32, 32, 3e-4), ("tiny_gqa", torch.float32, 48, 3e-4), ("llava15_7b", torch.bfloat16, 128, 3e-2),
                                            ("llava15_7b", torch.bfloat16, 2048, 3e-2)])      # the last: one 7B layer at model_max_length 2048 (scripts/finetune.sh)
def test_decoder_layer_backward_composes(cuda, name, dt, T, tol):
    from dataclasses import replace
    from llava_mi355x import ops
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    cfg = replace(synth.with_layers(synth.CONFIGS[name], 1, 1), init="unit")
    torch.manual_seed(5)
    shapes = {k: s for k, s in synth.tensor_shapes(cfg).items() if k.startswith("model.layers.0.")}
    w = {}
    for k, shp in shapes.items():
        t = (1 + 0.1 * torch.randn(shp)) if k.endswith("norm.weight") else torch.randn(shp) / math.sqrt(shp[-1])
        w[k] = t.to(dt).float().requires_grad_(True)              # values representable in the engine dtype; autograd in fp32
    H = cfg.hidden_size
    h = torch.randn(1, T, H).to(dt).float().requires_grad_(True)
    cos, sin = O.rope_cos_sin(cfg, torch.arange(T)[None], torch.float32)
    bias = torch.zeros(1, 1, T, T).masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None], torch.finfo(torch.float32).min)
    out, _ = O.decoder_layer(w, cfg, 0, h, cos, sin, None, bias)
    dout = torch.randn_like(out).to(dt).float()
    out.backward(dout)
    table = torch.from_numpy(O.rope_table(cfg, max(256, T))).to(cuda)
    wd = {k: v.detach() for k, v in w.items()}
    dh, grads = _layer_backward(ops, cfg, wd, 0, h.detach(), dout, dt, cuda, table)
    assert _rel(dh, h.grad[0]) <= tol, ("dh", _rel(dh, h.grad[0]))
    for k, g in grads.items():
        r = _rel(g, w[k].grad)
        assert r <= tol, (k, r)
