"""Split decode attention whose chunk merge happens in o_proj's staging (csrc/decode_flow.hip: flow_attn form 3 stops at the per-chunk partials;
csrc/gemm.hip: gemv2m_kernel merges them while its first weight rounds are on the wire) against the in-launch merge it replaces (LMX_ATTN_MERGE=0:
ticket merge by the last workgroup of a head, then the plain gemv2_kernel).  It is the decoder-attention half of LlamaAttention.forward for one new
token (HF5:models/llama/modeling_llama.py:243-288 via llava_llama.py:88-99) followed by o_proj; both forms run the same arithmetic in the same order, so
generated ids and logits must be BIT-IDENTICAL: head_dim 128 and 64 (GQA), bf16 and fp16, contexts that cross the 8-chunk (two template instances) and
16-chunk (falls back to the in-launch merge) limits, 7B and 13B widths (one and two 16-byte chunks of x per thread)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(cfg, dtype, merge, weights=None, **kw):
    from synthetic import build as harness
    new = {"LMX_ATTN_MERGE": "1" if merge else "0", "LMX_DECODE_FLOW": "0"}
    old = {k: os.environ.get(k) for k in new}
    os.environ.update(new)
    try:
        return harness.build_model(cfg, dtype=dtype, seed=0, weights=weights, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _request(cfg, cuda, dtype, length=24, seed=2):
    from synthetic import recipes as synth
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(5,), seed=seed))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=seed + 1)).to(cuda, dtype)
    return ids, pix


def _kernels(model, ids, pix):
    model.profile(True)
    model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=2, eos_token_id=-1)
    names = set(model.profile_read())
    model.profile(False)
    return names


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_merge_in_o_proj_is_bit_identical(cuda, name, dtype):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    a = _build(cfg, dtype, True, weights=wnp)
    b = _build(cfg, dtype, False, weights=wnp)
    ids, pix = _request(cfg, cuda, dtype)
    assert "decode.gemv.o" in _kernels(a, ids, pix)
    for run_ahead in (1, 6):
        ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1, run_ahead=run_ahead)
        gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1, run_ahead=run_ahead)
        assert torch.equal(ga, gb)
    oa = a.forward(input_ids=ids, images=pix); ob = b.forward(input_ids=ids, images=pix)
    tok = torch.tensor([[7]], device=cuda)
    assert torch.equal(a.forward(input_ids=tok, past_key_values=oa.past_key_values).logits, b.forward(input_ids=tok, past_key_values=ob.past_key_values).logits)


@pytest.mark.parametrize("name,start,new", [("tiny", 120, 20), ("tiny_gqa", 1015, 20), ("tiny", 2040, 16)])
def test_merge_across_chunk_limits(cuda, name, start, new):
    """Contexts 120 -> 140 (1 -> 2 live chunks), 1015 -> 1035 (8 -> 9: the 8-slot instance hands over to the 16-slot one) and 2040 -> 2056 (16 -> 17:
    the merge goes back into the attention launch) — the logits of every step, not only the picks."""
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    a = _build(cfg, torch.bfloat16, True, weights=wnp, max_position=4096)
    b = _build(cfg, torch.bfloat16, False, weights=wnp, max_position=4096)
    n_img = a.get_vision_tower().num_patches
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=start - n_img + 1)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=new, eos_token_id=-1, run_ahead=3)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=new, eos_token_id=-1, run_ahead=3)
    assert ga.shape[1] == ids.shape[1] + new
    assert torch.equal(ga, gb)
    oa = a.forward(input_ids=ids, images=pix); ob = b.forward(input_ids=ids, images=pix)
    pa, pb = oa.past_key_values, ob.past_key_values
    for t in range(new):
        tok = ga[:, ids.shape[1] + t: ids.shape[1] + t + 1]
        ra = a.forward(input_ids=tok, past_key_values=pa); rb = b.forward(input_ids=tok, past_key_values=pb)
        assert torch.equal(ra.logits, rb.logits), t
        pa, pb = ra.past_key_values, rb.past_key_values


@pytest.mark.parametrize("model,length", [("llava15_7b", 40), ("llava15_7b", 520), ("llava15_13b", 40), ("llava15_13b", 520)])
def test_merge_real_widths(cuda, model, length):
    """LLaVA-1.5-7B (K = 4096: one chunk of x per thread) and 13B (K = 5120: two) widths, 2 decoder layers; contexts ~600 (5 chunks) and ~1100 (9 chunks;
    13B: more than 8 chunks with two chunks of x per thread is not instantiated -> in-launch merge, still equal)."""
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS[model], 2, 1)
    a = _build(cfg, torch.bfloat16, True, device_rng=True, max_position=2048)
    b = _build(cfg, torch.bfloat16, False, device_rng=True, max_position=2048)
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=length)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1)
    assert torch.equal(ga, gb)
    oa = a.forward(input_ids=ids, images=pix); ob = b.forward(input_ids=ids, images=pix)
    tok = torch.tensor([[11]], device=cuda)
    assert torch.equal(a.forward(input_ids=tok, past_key_values=oa.past_key_values).logits, b.forward(input_ids=tok, past_key_values=ob.past_key_values).logits)
