"""RoPE + KV-cache append in the q|k|v GEMM's epilogue (csrc/gemm8p.hip: qkv_rope_epilogue) against the two launches it replaces
(launch_gemm + rope_kv_kernel).  Both are LlamaAttention's q_proj / k_proj / v_proj, apply_rotary_pos_emb and the cache update of a prefill
(HF5:models/llama/modeling_llama.py:130-160, 243-281) with the same arithmetic and the same rounding points (the tile is rounded to the model dtype
before the rotation, as a stored-and-reloaded q|k|v row would be), so logits and generated ids must be BIT-IDENTICAL — the ids also prove the caches:
every decode step reads the K / V^T rows the prefill wrote.  7B and 13B widths, prompts whose last row group is partial (1087 = 135 x 8 + 7 rows),
bf16 and fp16; a prompt too short for the ping-pong kernel and a chunk that starts at a position that is not a multiple of 8 keep the two launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(model, ids, pix, fuse, n_new=6, **kw):
    model.set_option("fuse_rope", 1 if fuse else 0)
    try:
        model.profile(True)
        out = model.forward(input_ids=ids, images=pix, use_cache=True)
        names = set(model.profile_read())
        model.profile(False)
        logits = out.logits.clone()
        out.past_key_values.close()
        gen = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=n_new, eos_token_id=-1, **kw)
        return logits, gen, names
    finally:
        model.set_option("fuse_rope", 1)


@pytest.mark.parametrize("name,dtype,length", [("llava15_7b", torch.bfloat16, 512), ("llava15_7b", torch.bfloat16, 500), ("llava15_7b", torch.float16, 512), ("llava15_7b", torch.float16, 500), ("llava15_13b", torch.bfloat16, 512)])
def test_fused_qkv_epilogue_is_bit_identical(cuda, name, dtype, length):
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS[name], 2, 1)
    model = harness.build_model(cfg, dtype=dtype, seed=0, device_rng=True, max_position=2048)
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(35,), seed=2))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(cuda, dtype)
    la, ga, na = _run(model, ids, pix, True)
    lb, gb, nb = _run(model, ids, pix, False)
    assert "prefill.rope_kv" not in na and "prefill.rope_kv" in nb, (na, nb)      # the fused form really ran, and really replaced the launch
    assert torch.equal(la, lb)
    assert torch.equal(ga, gb)


def test_shapes_that_keep_the_two_launches(cuda):
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, max_position=2048)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(cuda, torch.bfloat16)
    # 40 + 575 rows: too few 256 x 256 tiles for the ping-pong kernel
    ids = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(5,), seed=3))[None].to(cuda)
    _, _, names = _run(model, ids, pix, True, n_new=2)
    assert "prefill.rope_kv" in names
    # chunks of 1000 rows: the second chunk starts at position 1000 (a multiple of 8: fused), chunks of 1001 would not — both must equal the one-shot prefill
    ids = torch.from_numpy(synth.make_prompt(cfg, 1200, image_positions=(35,), seed=4))[None].to(cuda)
    one = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=4, eos_token_id=-1)
    for chunk in (1000, 1001):
        got = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=4, eos_token_id=-1, prefill_chunk=chunk)
        assert torch.equal(got, one), chunk


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,nh,nkv,T,pos0", [(128, 32, 32, 1087, 0), (128, 16, 16, 2000, 40), (64, 64, 16, 2000, 8)])
def test_fused_projection_op_equals_gemm_then_rope_kv(cuda, dtype, D, nh, nkv, T, pos0):
    """The two forms on the same operands: rotated q, the K-cache rows and the V^T columns must be the same BITS (GQA head_dim 64 included: four heads per
    tile, rotation partner inside the wave's own 64 columns)."""
    from llava_mi355x import ops
    from test_ops_gpu import _rope_table
    torch.manual_seed(3)
    K, s_max = 1024, 2304
    N = (nh + 2 * nkv) * D
    x = (torch.randn(T, K, device=cuda) * 0.5).to(dtype)
    w = (torch.randn(N, K, device=cuda) * 0.05).to(dtype)
    table = _rope_table(s_max, D).to(cuda)
    ref = ops.gemm(x, w)
    k0, v0 = ops.alloc_kv(nkv, s_max, D, dtype)
    ops.rope_kv(ref, k0, v0, table, pos0, nh, nkv, D)
    got = torch.full((T, N), 7.0, device=cuda, dtype=dtype)
    k1, v1 = ops.alloc_kv(nkv, s_max, D, dtype)
    ops.gemm_qkv_rope(x, w, got, k1, v1, table, pos0, nh, nkv, D)
    bits = lambda t: t.view(torch.int16)
    dq = (bits(got[:, : nh * D]) != bits(ref[:, : nh * D])).sum().item()
    dk = (bits(k1) != bits(k0)).sum().item()
    dv = (bits(v1) != bits(v0)).sum().item()
    assert (dq, dk, dv) == (0, 0, 0), (dq, dk, dv)
    assert torch.all(got[:, nh * D:] == 7.0)                       # the k | v columns of the output buffer are not written
