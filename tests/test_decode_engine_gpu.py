"""Persistent decode step with data-tagged hand-overs (csrc/decode_engine.hip: one launch per token, 4 resident workgroups per CU, activation rows as {pair, tag} granules)
against the separate launches it replaces (Model::decode_step_launch's GEMV / fused attention kernels; the decoder half of LlamaModel.forward for one
new token, HF5:models/llama/modeling_llama.py:367-418 via llava_llama.py:88-99).  Both run the same arithmetic in the same order, so generated ids
and logits must be BIT-IDENTICAL, in bf16 and fp16, for head_dim 128 and 64 (GQA), at the tiny geometries and at real LLaVA-1.5-7B widths, across
a 128-key chunk boundary of the KV cache, with several request threads sharing the chip, and for sampled requests.  fp32 models keep the separate
launches (and their parity with the oracle is tested elsewhere); every test asserts which path actually ran through the in-situ profiler."""
import os
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(cfg, dtype, flow, weights=None, env=None, **kw):
    from synthetic import build as harness
    new = {"LMX_DECODE_ENGINE": "1" if flow else "0", "LMX_DECODE_FLOW": "0"}
    new.update(env or {})
    old = {k: os.environ.get(k) for k in new}
    os.environ.update(new)
    try:
        model = harness.build_model(cfg, dtype=dtype, seed=0, weights=weights, **kw)
        # the switches are read at the first decode step of a model: take them now, and check which path runs
        ids = torch.tensor([[1, 5, 7, 9]], device=model.device)
        model.profile(True)
        model.generate(inputs=ids, do_sample=False, max_new_tokens=2, eos_token_id=-1)
        names = set(model.profile_read())
        model.profile(False)
        if flow and dtype != torch.float32:
            assert "decode.engine" in names and "decode.gemv.qkv" not in names, names
        else:
            assert "decode.engine" not in names and "decode.gemv.qkv" in names, names
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return model


def _request(cfg, cuda, dtype, length=24, seed=2):
    from synthetic import recipes as synth
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(5,), seed=seed))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=seed + 1)).to(cuda, dtype)
    return ids, pix


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_engine_step_is_bit_identical(cuda, name, dtype):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    a = _build(cfg, dtype, flow=True, weights=wnp)
    b = _build(cfg, dtype, flow=False, weights=wnp)
    ids, pix = _request(cfg, cuda, dtype)
    for run_ahead in (1, 7):
        ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1, run_ahead=run_ahead)
        gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=20, eos_token_id=-1, run_ahead=run_ahead)
        assert torch.equal(ga, gb)
    # logits of a decode step through forward(past_key_values=...)
    oa = a.forward(input_ids=ids, images=pix); ob = b.forward(input_ids=ids, images=pix)
    tok = torch.tensor([[7]], device=cuda)
    la = a.forward(input_ids=tok, past_key_values=oa.past_key_values).logits
    lb = b.forward(input_ids=tok, past_key_values=ob.past_key_values).logits
    assert torch.equal(la, lb)
    # sampled: the draw kernel runs on the flow kernel's logits
    torch.manual_seed(5); sa = a.generate(inputs=ids, images=pix, do_sample=True, temperature=0.8, top_p=0.9, max_new_tokens=12, eos_token_id=-1)
    torch.manual_seed(5); sb = b.generate(inputs=ids, images=pix, do_sample=True, temperature=0.8, top_p=0.9, max_new_tokens=12, eos_token_id=-1)
    assert torch.equal(sa, sb)


def test_engine_step_across_a_chunk_boundary(cuda):
    """Contexts 120 -> 140 cross the first 128-key chunk boundary: the number of attention workgroups (live chunks) changes from one launch to the
    next, a chunk that holds only the newest key appears, and the per-head tickets / completion counters must follow."""
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    a = _build(cfg, torch.bfloat16, flow=True, weights=wnp)
    b = _build(cfg, torch.bfloat16, flow=False, weights=wnp)
    n_img = a.get_vision_tower().num_patches
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=121 - n_img + 1)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=40, eos_token_id=-1, run_ahead=3)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=40, eos_token_id=-1, run_ahead=3)
    assert ga.shape[1] == ids.shape[1] + 40
    assert torch.equal(ga, gb)


def test_engine_step_real_widths(cuda):
    """LLaVA-1.5-7B widths (H 4096, I 11008, 32 heads x 128, V 32000), 2 decoder layers, context ~600 -> 5 live chunks per head."""
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    a = _build(cfg, torch.bfloat16, flow=True, device_rng=True, max_position=1024)
    b = _build(cfg, torch.bfloat16, flow=False, device_rng=True, max_position=1024)
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=40)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1)
    assert torch.equal(ga, gb)
    oa = a.forward(input_ids=ids, images=pix); ob = b.forward(input_ids=ids, images=pix)
    tok = torch.tensor([[11]], device=cuda)
    assert torch.equal(a.forward(input_ids=tok, past_key_values=oa.past_key_values).logits, b.forward(input_ids=tok, past_key_values=ob.past_key_values).logits)


def test_engine_step_13b_widths(cuda):
    """LLaVA-1.5-13B widths (H 5120, I 13824, 40 heads): the RMSNorm staging sweeps a row that is not a power of two."""
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_13b"], 2, 1)
    a = _build(cfg, torch.bfloat16, flow=True, device_rng=True, max_position=1024)
    b = _build(cfg, torch.bfloat16, flow=False, device_rng=True, max_position=1024)
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=40)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=16, eos_token_id=-1)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=16, eos_token_id=-1)
    assert torch.equal(ga, gb)


def test_fp32_model_keeps_separate_launches(cuda):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    m = _build(cfg, torch.float32, flow=True, weights=synth.make_weights(cfg, 0))
    ids, pix = _request(cfg, cuda, torch.float32)
    out = m.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=6, eos_token_id=-1)
    assert out.shape[1] == ids.shape[1] + 6


def test_request_threads_share_the_chip(cuda):
    """The worker's thread-per-request model (model_worker.py:174-185): several requests on different streams.  A persistent grid needs every workgroup resident, so launches
    of different sequences queue behind each other on the device (the host chains them with an event); ids must equal the one-at-a-time ids."""
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    m = _build(cfg, torch.bfloat16, flow=True, weights=synth.make_weights(cfg, 0))
    reqs = [_request(cfg, cuda, torch.bfloat16, length=20 + 3 * i, seed=10 + i) for i in range(4)]
    want = [m.generate(inputs=i, images=p, do_sample=False, max_new_tokens=30, eos_token_id=-1, run_ahead=4) for i, p in reqs]
    got = [None] * len(reqs)

    def run(k):
        with torch.cuda.stream(torch.cuda.Stream()):
            got[k] = m.generate(inputs=reqs[k][0], images=reqs[k][1], do_sample=False, max_new_tokens=30, eos_token_id=-1, run_ahead=4)
            torch.cuda.current_stream().synchronize()

    ths = [threading.Thread(target=run, args=(k,)) for k in range(len(reqs))]
    for t in ths: t.start()
    for t in ths: t.join(timeout=120)
    assert all(not t.is_alive() for t in ths)
    for g, w in zip(got, want):
        assert g is not None and torch.equal(g.cpu(), w.cpu())


def test_request_threads_share_the_chip_real_widths(cuda):
    """Same, with grids that fill the chip (7B widths, 2 layers): two sequences' workgroups interleave on every CU."""
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    m = _build(cfg, torch.bfloat16, flow=True, device_rng=True, max_position=1024)
    reqs = [_request(cfg, cuda, torch.bfloat16, length=30 + 5 * i, seed=20 + i) for i in range(3)]
    want = [m.generate(inputs=i, images=p, do_sample=False, max_new_tokens=16, eos_token_id=-1, run_ahead=8) for i, p in reqs]
    got = [None] * len(reqs)

    def run(k):
        with torch.cuda.stream(torch.cuda.Stream()):
            got[k] = m.generate(inputs=reqs[k][0], images=reqs[k][1], do_sample=False, max_new_tokens=16, eos_token_id=-1, run_ahead=8)
            torch.cuda.current_stream().synchronize()

    ths = [threading.Thread(target=run, args=(k,)) for k in range(len(reqs))]
    for t in ths: t.start()
    for t in ths: t.join(timeout=120)
    assert all(not t.is_alive() for t in ths)
    for g, w in zip(got, want):
        assert g is not None and torch.equal(g.cpu(), w.cpu())
