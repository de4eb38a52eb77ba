"""Persistent decode-step kernel (csrc/decode_persist.hip) against the separate launches it replaces (Model::decode_step_launch's GEMV / fused
attention kernels).  Both run the same arithmetic in the same order — the linear phases reproduce gemv_kernel's per-lane accumulation, the attention
phase IS decode_fused_kernel's code — so generated ids and logits must be bit-identical, in bf16 and fp16, for head_dim 128 and 64 (GQA), at the
tiny geometries and at real LLaVA-1.5-7B widths; fp32 models and tensor-parallel models keep the separate launches.  Also: two request threads on
two streams (the worker's thread-per-request model) must not deadlock the co-resident grids, and a sampled request must draw the same ids."""
import os
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(cfg, dtype, persist, weights=None, **kw):
    from synthetic import build as harness
    old = os.environ.get("LMX_DECODE_PERSIST")
    os.environ["LMX_DECODE_PERSIST"] = "1" if persist else "0"
    try:
        model = harness.build_model(cfg, dtype=dtype, seed=0, weights=weights, **kw)
        # the switch is read at the first decode step of a model: take it now
        ids = torch.tensor([[1, 5, 7, 9]], device=model.device)
        model.generate(inputs=ids, do_sample=False, max_new_tokens=2, eos_token_id=-1)
    finally:
        if old is None:
            os.environ.pop("LMX_DECODE_PERSIST", None)
        else:
            os.environ["LMX_DECODE_PERSIST"] = old
    return model


def _request(cfg, cuda, dtype, length=24, seed=2):
    from synthetic import recipes as synth
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(5,), seed=seed))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=seed + 1)).to(cuda, dtype)
    return ids, pix


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_persistent_step_is_bit_identical(cuda, name, dtype):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    a = _build(cfg, dtype, persist=True, weights=wnp)
    b = _build(cfg, dtype, persist=False, weights=wnp)
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
    # sampled: the draw kernel runs on the persistent kernel's logits
    torch.manual_seed(5); sa = a.generate(inputs=ids, images=pix, do_sample=True, temperature=0.8, top_p=0.9, max_new_tokens=12, eos_token_id=-1)
    torch.manual_seed(5); sb = b.generate(inputs=ids, images=pix, do_sample=True, temperature=0.8, top_p=0.9, max_new_tokens=12, eos_token_id=-1)
    assert torch.equal(sa, sb)


def test_persistent_step_real_widths(cuda):
    """LLaVA-1.5-7B widths (H 4096, I 11008, 32 heads x 128, V 32000), 2 decoder layers, context ~600."""
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    a = _build(cfg, torch.bfloat16, persist=True, device_rng=True, max_position=1024)
    b = _build(cfg, torch.bfloat16, persist=False, device_rng=True, max_position=1024)
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=40)
    ga = a.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1)
    gb = b.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1)
    assert torch.equal(ga, gb)


def test_fp32_model_keeps_separate_launches(cuda):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    m = _build(cfg, torch.float32, persist=True, weights=synth.make_weights(cfg, 0))
    ids, pix = _request(cfg, cuda, torch.float32)
    out = m.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=6, eos_token_id=-1)
    assert out.shape[1] == ids.shape[1] + 6


def test_two_request_threads_do_not_deadlock(cuda):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    m = _build(cfg, torch.bfloat16, persist=True, weights=synth.make_weights(cfg, 0))
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


def test_stopping_criteria_with_run_ahead_stop_where_the_per_token_check_stops(cuda):
    """A request with stopping criteria but no streamer chains decode steps on the device and checks the criteria per prefix on the ids it reads
    back: the returned ids must be those of the token-by-token loop (reference: KeywordsStoppingCriteria is asked after every token,
    llava/mm_utils.py:94-107; model_worker.py:174-185)."""
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    m = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, weights=synth.make_weights(cfg, 0))
    ids, pix = _request(cfg, cuda, torch.bfloat16)
    free = m.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1)[0, ids.shape[1]:].tolist()
    calls = []

    def stop_on_pair(full, scores):                     # stop once the 9th and 10th free-running tokens have just been produced
        calls.append(full.shape[1])
        return full[0, -2:].tolist() == free[8:10]

    outs = []
    for ahead in (1, 5, 16):
        calls.clear()
        o = m.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1, stopping_criteria=[stop_on_pair], run_ahead=ahead)
        outs.append(o[0, ids.shape[1]:].tolist())
        assert calls == list(range(ids.shape[1] + 1, ids.shape[1] + 11))          # asked after every token, in order, and not past the stop
    assert outs[0] == free[:10] and outs[1] == outs[0] and outs[2] == outs[0]
