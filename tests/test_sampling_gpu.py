"""Device sampler (csrc/sampling.hip) against the reference's sampling semantics: the survivor set must be the one HF's own
TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper leave (the chain model.generate(do_sample=True, temperature,
top_p) builds for llava/serve/model_worker.py:156-184), and the drawn id must be the inverse-CDF pick over those survivors
for the uniform number used."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _device_sample(logits, T, top_p, top_k, u32=None, seed=0, offset=None, want_keep=True):
    from llava_mi355x import _C
    V = logits.numel()
    out = torch.zeros(1, dtype=torch.long, device=logits.device)
    keep = torch.zeros(V, dtype=torch.uint8, device=logits.device) if want_keep else None
    uo = ctypes.pointer(ctypes.c_uint32(int(u32))) if u32 is not None else None
    _C.check(_C.lib.lmx_op_sample(_C.torch_dtype_code(logits.dtype), _C.ptr(logits), V, float(T), float(top_p), int(top_k), int(seed),
                                  _C.ptr(offset), uo, _C.ptr(out), _C.ptr(keep), _C.stream_handle()), "lmx_op_sample")
    torch.cuda.synchronize()
    return int(out.item()), (keep.cpu().numpy().astype(bool) if want_keep else None)


def _hf_keep(logits_f32_cpu, T, top_p, top_k):
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    s = logits_f32_cpu[None].clone()
    ids = torch.zeros((1, 1), dtype=torch.long)
    s = TemperatureLogitsWarper(float(T))(ids, s)
    if top_k:
        s = TopKLogitsWarper(int(top_k))(ids, s)
    if top_p < 1.0:
        s = TopPLogitsWarper(float(top_p))(ids, s)
    return torch.isfinite(s[0]).numpy(), s[0]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [1000, 32000])
@pytest.mark.parametrize("T,top_p,top_k", [(1.0, 1.0, 0), (0.7, 0.9, 0), (1.0, 0.5, 0), (0.2, 0.7, 0), (1.3, 1.0, 50), (0.8, 0.95, 40), (1.0, 0.3, 5), (2.0, 0.99, 0)])
def test_survivor_set_and_draw_match_reference_warpers(cuda, dt, V, T, top_p, top_k):
    g = torch.Generator().manual_seed(V + int(T * 10) + top_k)
    logits = (torch.randn(V, generator=g) * 3.0).to(dt)
    ref_keep, ref_scores = _hf_keep(logits.float(), T, top_p, top_k)
    tok, keep = _device_sample(logits.cuda(), T, top_p, top_k, u32=123456789)
    diff = np.flatnonzero(keep != ref_keep)
    # Differences are only allowed AT the nucleus boundary: (a) tokens tied with the boundary logit (bf16 logits tie often; the
    # reference's unstable sort keeps an arbitrary subset of a tie class, this kernel keeps the whole class), (b) one token whose
    # cumulative mass sits on the threshold (fp32 exp on the device vs torch's softmax).
    lf = logits.float().numpy()
    if len(diff):
        boundary = min(lf[keep].min(), lf[ref_keep].min())
        tied = np.isclose(lf[diff], boundary, rtol=0, atol=0)
        assert (~tied).sum() <= 1, (diff, lf[diff], boundary)
        assert np.all(lf[diff] <= np.partition(lf[keep], 1)[1] + 1e-6)           # never a token above the boundary class
    assert keep[int(torch.argmax(logits.float()))]
    # the draw: inverse CDF in token-id order over the survivors, float64
    e = np.exp((logits.float().numpy().astype(np.float64) - float(logits.float().max())) / T) * keep
    cdf = np.cumsum(e)
    for u32 in (0, 1, 123456789, 2 ** 31, 2 ** 32 - 1, 987654321, 3141592653):
        tok, _ = _device_sample(logits.cuda(), T, top_p, top_k, u32=u32, want_keep=False)
        assert keep[tok]
        target = (u32 / 2.0 ** 32) * cdf[-1]
        lo = cdf[tok - 1] if tok > 0 else 0.0
        tol = 1e-4 * cdf[-1]          # masses are 31-bit fixed point on the device: up to V * 2^-31 of drift along the CDF
        assert lo - tol <= target <= cdf[tok] + tol, (u32, tok, lo, target, cdf[tok])


def test_philox_draws_follow_the_distribution(cuda):
    """4000 draws (counter = 0..3999) from a 6-token distribution: frequencies within 5 sigma of the probabilities; same seed and
    counter -> same id; another seed -> another sequence."""
    logits = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, 1.5], device=cuda)
    p = torch.softmax(logits.float() / 0.9, dim=0).cpu().numpy()
    n = 4000
    off = torch.zeros(1, dtype=torch.int32, device=cuda)
    seq_a, seq_b = [], []
    for i in range(n):
        off.fill_(i)
        seq_a.append(_device_sample(logits, 0.9, 1.0, 0, seed=42, offset=off, want_keep=False)[0])
        if i < 64:
            seq_b.append(_device_sample(logits, 0.9, 1.0, 0, seed=43, offset=off, want_keep=False)[0])
    off.fill_(7)
    assert _device_sample(logits, 0.9, 1.0, 0, seed=42, offset=off, want_keep=False)[0] == seq_a[7]
    assert seq_a[:64] != seq_b
    freq = np.bincount(seq_a, minlength=6) / n
    assert np.all(np.abs(freq - p) <= 5 * np.sqrt(p * (1 - p) / n) + 1e-3), (freq, p)


def test_generate_sampling_on_device(cuda):
    """generate(do_sample=True): reproducible under torch.manual_seed, different under another seed, chained steps (run_ahead) and the
    continuous-batching scheduler give the same ids as step-by-step for the same seed (fp32 engine), temperature -> 0 approaches greedy."""
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    ids = torch.from_numpy(synth.make_prompt(cfg, 12, image_positions=(3,)))[None].cuda()
    pix = torch.from_numpy(synth.make_pixels(cfg, 1)).cuda()

    def gen(seed, **kw):
        torch.manual_seed(seed)
        return model.generate(inputs=ids, images=pix, do_sample=True, temperature=kw.pop("temperature", 1.0), top_p=kw.pop("top_p", 0.95),
                              max_new_tokens=16, eos_token_id=-1, **kw).cpu()

    a, b, c = gen(1), gen(1), gen(2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(a, gen(1, run_ahead=1)) and torch.equal(a, gen(1, run_ahead=5))
    greedy = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=16, eos_token_id=-1).cpu()
    assert torch.equal(gen(3, temperature=0.01, top_p=1.0), greedy)
    assert torch.equal(gen(4, top_p=1e-6), greedy)                # the nucleus always keeps the most likely token
    model.enable_batching(capacity=4)
    try:
        assert torch.equal(a, gen(1))
    finally:
        model.disable_batching()
