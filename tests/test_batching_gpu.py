"""Decode batch / continuous batching: the skinny MFMA linear, lmx_decode_batch and the scheduler.

Oracle statement: batching must not change what any request computes — a sequence stepped inside a batch (own KV cache, own
position, neighbours of other lengths joining and leaving) produces the ids the reference produces for that request alone
(tests/golden: `generate`), and the logits of the single-sequence engine within dtype tolerance."""
import ctypes
import math
import threading

import numpy as np
import pytest
import torch

from golden_util import case_inputs, load

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
SKINNY = 20


def _rel_err(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(1, 256, 128), (3, 4096, 4096), (8, 1024, 11008), (16, 12288, 4096), (17, 4096, 1408),
                                   (32, 1000 // 8 * 8 + 8, 1728), (5, 32000, 4096), (32, 4096, 13824),
                                   (32, 12288, 1024), (20, 32000, 512)])       # > 16 rows and N >= 8192: the 64-rows-per-workgroup variant
def test_skinny_gemm_plain(cuda, dt, M, N, K):
    from llava_mi355x import ops
    torch.manual_seed(M * 31 + N)
    x = torch.randn(M, K, device=cuda).to(DT[dt]); w = (torch.randn(N, K, device=cuda) / math.sqrt(K)).to(DT[dt])
    got = ops.gemm(x, w, variant=SKINNY)
    ref = x.float() @ w.float().t()
    assert _rel_err(got, ref) < 1e-2
    # same kernel reading the fragment-order copy of w (what the engine's decode batch uses): identical bits
    got_sw = ops.gemm(x, w, variant=SKINNY + 1)
    assert torch.equal(got, got_sw)


def test_skinny_gemm_is_transpose_detecting(cuda):
    """identity-like x against an asymmetric W: exact, catches any row/col/k-permutation slip in the MFMA operand mapping."""
    from llava_mi355x import ops
    M, N, K = 16, 64, 256
    x = torch.zeros(M, K, device=cuda, dtype=torch.bfloat16)
    for t in range(M):
        x[t, (t * 37 + 5) % K] = 1.0; x[t, (t * 11 + 130) % K] = 2.0
    w = (torch.arange(N, device=cuda).float()[:, None] * 0.25 + torch.arange(K, device=cuda).float()[None, :] * 0.001953125).to(torch.bfloat16)
    got = ops.gemm(x, w, variant=SKINNY)
    assert torch.equal(got.float().cpu(), (x.float() @ w.float().t()).to(torch.bfloat16).float().cpu())


@pytest.mark.parametrize("M", [2, 16, 29])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_skinny_gemm_bias_act_residual(cuda, M, act):
    from llava_mi355x import ops
    torch.manual_seed(act + M)
    N, K = 272, 192
    x = torch.randn(M, K, device=cuda).bfloat16(); w = (torch.randn(N, K, device=cuda) / math.sqrt(K)).bfloat16()
    b = torch.randn(N, device=cuda).bfloat16(); r = torch.randn(M, N, device=cuda).bfloat16()
    pre = x.float() @ w.float().t() + b.float()
    ref = (pre if act == 0 else pre * torch.sigmoid(1.702 * pre) if act == 1 else torch.nn.functional.gelu(pre)) + r.float()
    got = ops.gemm(x, w, bias=b, residual=r, act=act, variant=SKINNY)
    assert _rel_err(got, ref) < 1e-2
    r2 = r.clone()
    ops.gemm(x, w, bias=b, residual=r2, act=act, out=r2, variant=SKINNY)      # C aliases R (the residual stream)
    assert _rel_err(r2, ref) < 1e-2


@pytest.mark.parametrize("M,I", [(1, 352), (7, 352), (16, 352), (32, 352), (20, 8192), (32, 11008)])
def test_skinny_gemm_silu_mul(cuda, M, I):
    from llava_mi355x import _C, ops
    torch.manual_seed(5 + M)
    K = 256
    x = torch.randn(M, K, device=cuda).bfloat16()
    g = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16(); u = (torch.randn(I, K, device=cuda) / math.sqrt(K)).bfloat16()
    fused = ops.interleave_gate_up(g, u)
    ref = torch.nn.functional.silu(x.float() @ g.float().t()) * (x.float() @ u.float().t())
    got = ops.gemm(x, fused, act=_C.ACT_SILU_MUL, variant=SKINNY)
    assert got.shape == (M, I) and _rel_err(got, ref) < 1e-2
    assert torch.equal(got, ops.gemm(x, fused, act=_C.ACT_SILU_MUL, variant=SKINNY + 1))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(9, 4096, 4096), (16, 4096, 4096), (17, 4096, 11008), (32, 4096, 4096), (32, 4096, 11008), (32, 5120, 13824), (24, 4096, 1408),
                                   (32, 4096, 512), (12, 1024, 4096), (32, 4160, 4096),
                                   (32, 12288, 4096), (17, 12288, 4096), (24, 15360, 5120), (16, 12288, 4096)])     # q|k|v of 7B / 13B: two slices above 16 rows
def test_skinny_gemm_k_slices_across_workgroups(cuda, dt, M, N, K):
    """Round 6: narrow layers (N = hidden) of a decode batch with more than 8 rows take 64 weight rows per workgroup and K SLICES ACROSS WORKGROUPS (variant 24: the
    engine's form, with its scratch): every slice's fp32 partial tile leaves write-through, the last workgroup to arrive for a tile adds the slices in slice order and
    runs the epilogue.  Checked against the float64 product rounded once (one ulp + fp32 accumulation noise), against the 16-rows-per-workgroup form (same tolerance:
    the summation order differs), bit for bit from launch to launch on the SAME scratch and tickets (30 launches, alternating inputs: a stale partial tile or a ticket
    left armed would show), with bias / activation / residual aliasing the output."""
    from llava_mi355x import ops
    from test_gemm8p_gpu import assert_one_ulp
    g = torch.Generator(device=cuda).manual_seed(M * 131 + N + K)
    T = DT[dt]
    w = (torch.randn(N, K, device=cuda, generator=g) / math.sqrt(K)).to(T)
    xs = [torch.randn(M, K, device=cuda, generator=g).to(T) for _ in range(2)]
    refs = [x.double() @ w.double().t() for x in xs]
    first = [None, None]
    for it in range(30):
        i = it & 1
        got = ops.gemm(xs[i], w, variant=24)
        if first[i] is None:
            first[i] = got.clone()
            assert_one_ulp(got, refs[i], dt, 3e-5 * float(refs[i].abs().max()), f"K-slice form, input {i}")
            assert_one_ulp(ops.gemm(xs[i], w, variant=21), refs[i], dt, 3e-5 * float(refs[i].abs().max()), "16-rows-per-workgroup form")
        else:
            assert torch.equal(got, first[i]), f"launch {it}: not the bits of the first launch on this input"
    b = torch.randn(N, device=cuda, generator=g).to(T); r = torch.randn(M, N, device=cuda, generator=g).to(T)
    pre = refs[0] + b.double()
    ref = pre * torch.sigmoid(1.702 * pre) + r.double()
    r2 = r.clone()
    ops.gemm(xs[0], w, bias=b, residual=r2, act=1, out=r2, variant=24)                 # C aliases R (the residual stream)
    assert_one_ulp(r2, ref, dt, 6e-5 * float(pre.abs().max()), "bias + quick_gelu + residual in place")


def _requests(cfg, n, seed0=100):
    """n requests of different prompt lengths, each with its own image."""
    from synthetic import recipes as synth
    reqs = []
    for i in range(n):
        L = 9 + 5 * (i % 4)
        ids = torch.from_numpy(synth.make_prompt(cfg, L, image_positions=(2 + i % 3,), seed=seed0 + i))
        pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=seed0 + 50 + i))
        reqs.append((ids, pix))
    return reqs


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
def test_batched_decode_fp32_ids_equal_single_and_oracle(cuda, name):
    """fp32 engine: a batch of 5 requests (different prompt lengths and images) decoded together gives, per request, exactly
    the ids of the request decoded alone AND of the oracle (CPU restatement of the reference) for that request."""
    from oracle import llava_oracle as O
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    reqs = _requests(cfg, 5)
    new = 7
    outs = model.generate_batch([r[0].cuda() for r in reqs], [r[1].cuda() for r in reqs], max_new_tokens=new, eos_token_id=-1, run_ahead=3)
    w = O.to_torch_weights(synth.make_weights(cfg, 0))
    for (ids, pix), o in zip(reqs, outs):
        single = model.generate(inputs=ids[None].cuda(), images=pix.cuda(), do_sample=False, max_new_tokens=new, eos_token_id=-1)
        assert torch.equal(o, single[0])
        with torch.no_grad():
            ref = O.greedy_generate(w, cfg, ids[None], pix, new)
        assert o[ids.shape[0]:].tolist() == ref


def test_batched_decode_matches_reference_golden(cuda):
    """the golden request (ids produced by the shimmed reference itself) placed in a batch among strangers."""
    from synthetic import build as harness
    z, meta = load("tiny")
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, "single")
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    others = _requests(cfg, 3, seed0=300)
    gold = z["single.generate"]
    new = gold.shape[1] - ids.shape[1]
    prompts = [others[0][0].cuda(), torch.from_numpy(ids[0]).cuda(), others[1][0].cuda(), others[2][0].cuda()]
    images = [others[0][1].cuda(), torch.from_numpy(pix).cuda(), others[1][1].cuda(), others[2][1].cuda()]
    outs = model.generate_batch(prompts, images, max_new_tokens=new, eos_token_id=-1, run_ahead=4)
    assert np.array_equal(outs[1].cpu().numpy(), gold[0])


@pytest.mark.parametrize("n_req", [2, 6])          # 2 members: multi-row GEMV chain; more: rmsnorm + skinny MFMA linears
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_batched_step_logits_match_single_step(cuda, dt, n_req):
    """16-bit engine: logits of one batched step (skinny MFMA linears, batched fused attention) vs the same step through the
    single-sequence path (GEMV + fused attention) for every member: <= 3e-2 of max|logit|; KV caches advance identically."""
    from llava_mi355x import _C
    from llava_mi355x.batching import DecodeBatch
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=dt, seed=0)
    reqs = _requests(cfg, n_req)
    V = cfg.vocab_size

    def prefilled():
        caches = []
        for ids, pix in reqs:
            _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids[None].cuda(), None, None, None, None, pix.cuda().to(dt))
            c = LmxKVCache(model, 1)
            _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
            caches.append(c)
        return caches

    a, b = prefilled(), prefilled()
    batch = DecodeBatch(model, 8)
    lb = torch.empty((len(reqs), V), dtype=dt, device=cuda)
    for step in range(3):
        batch.step([c.seqs[0] for c in a], None, 1, True, lb)
        for i, c in enumerate(b):
            ls = torch.empty((1, V), dtype=dt, device=cuda)
            # feed the batched path's previous pick so both paths see the same token even if a near-tie flips an argmax
            _C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, 1, _C.ptr(ls), 1, _C.stream_handle()))
            torch.cuda.synchronize()
            err = (lb[i].float() - ls[0].float()).abs().max().item() / ls.float().abs().max().item()
            assert err <= 3e-2, f"step {step} member {i}: {err:.3e}"
        assert [c.lengths()[0] for c in a] == [c.lengths()[0] for c in b]
    batch.close()
    for c in a + b:
        c.close()


def test_decode_batch_argument_errors(cuda):
    from llava_mi355x import _C
    from llava_mi355x.batching import DecodeBatch
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    batch = DecodeBatch(model, 2)
    c = LmxKVCache(model, 3)
    with pytest.raises(_C.LmxError, match="before prefill"):
        batch.step([c.seqs[0]])
    with pytest.raises(_C.LmxError, match="capacity"):
        batch.step(c.seqs)
    ids, pix = _requests(cfg, 1)[0]
    _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids[None].cuda(), None, None, None, None, pix.cuda())
    _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
    with pytest.raises(_C.LmxError, match="twice"):
        batch.step([c.seqs[0], c.seqs[0]])
    with pytest.raises(_C.LmxError, match="capacity"):
        batch.step([c.seqs[0]], n_steps=100000)
    with pytest.raises(_C.LmxError, match="out of range"):
        batch.step([c.seqs[0]], tokens=[cfg.vocab_size + 5])
    batch.close(); c.close()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_continuous_batching_scheduler(cuda, dt):
    """model_worker's threading model (one generate() thread per request, llava/serve/model_worker.py:174-185) with the
    scheduler on: requests of different lengths and budgets start at different times, join the running batch after their own
    prefill and leave at their own stop.  fp32: every request returns exactly what it returns alone; bf16: same lengths, and
    the scheduler really batched (member-steps > steps)."""
    import time
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=dt, seed=0)
    reqs = _requests(cfg, 7, seed0=500)
    budgets = [12, 5, 9, 12, 3, 8, 10]
    alone = [model.generate(inputs=ids[None].cuda(), images=pix.cuda().to(dt), do_sample=False, max_new_tokens=b, eos_token_id=-1)
             for (ids, pix), b in zip(reqs, budgets)]
    model.enable_batching(capacity=4)          # fewer slots than requests: some wait for a leaver
    model._batcher.pause()                     # line the first requests up so the first step is certainly a batched one
    results, errors = [None] * len(reqs), []

    class StopAt:                               # a stopping criterion, like KeywordsStoppingCriteria (mm_utils.py:79-114)
        def __init__(self, n): self.n = n
        def __call__(self, ids, scores, **kw): return ids.shape[1] >= self.n

    def run(i):
        try:
            time.sleep(0.002 * i)
            ids, pix = reqs[i]
            with torch.cuda.stream(torch.cuda.Stream()):
                kw = {}
                if i == 2:
                    kw["stopping_criteria"] = [StopAt(ids.shape[0] + 4)]
                results[i] = model.generate(inputs=ids[None].cuda(), images=pix.cuda().to(dt), do_sample=False,
                                            max_new_tokens=budgets[i], eos_token_id=-1, **kw).cpu()
        except Exception as e:  # noqa: BLE001
            errors.append((i, e))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(reqs))]
    for t in ths: t.start()
    t0 = time.time()
    while model._batcher.queued() < 4 and time.time() - t0 < 60:
        time.sleep(0.001)
    model._batcher.resume()
    for t in ths: t.join()
    stats = (model._batcher.steps, model._batcher.member_steps, model._batcher.max_live)
    model.disable_batching()
    assert not errors, errors
    for i, r in enumerate(results):
        want = alone[i].cpu()
        if i == 2:
            want = want[:, : reqs[i][0].shape[0] + 4]
        assert r.shape == want.shape
        if dt == torch.float32:
            assert torch.equal(r, want), f"request {i}"
    assert stats[2] >= 2 and stats[1] > stats[0], stats
    # sampled requests go through the same scheduler (host draw from the batch's logits rows)
    model.enable_batching(capacity=4)
    torch.manual_seed(0)
    out = model.generate(inputs=reqs[0][0][None].cuda(), images=reqs[0][1].cuda().to(dt), do_sample=True, temperature=0.8, top_p=0.9,
                         max_new_tokens=6, eos_token_id=-1)
    model.disable_batching()
    assert out.shape[1] == reqs[0][0].shape[0] + 6


@pytest.mark.parametrize("cname", ["batch_mixed", "batch_left_pad"])
def test_batched_generate_rows_share_decode_steps(cuda, cname):
    """model.generate with B > 1 (padded rows, one shared `images` argument: row b owns the next max(1, #<image>) entries — the slot
    arithmetic of llava_arch.py:150-159) decodes all rows together; every row must equal the same row generated alone, and the
    oracle's greedy continuation of that row."""
    from oracle import llava_oracle as O
    from synthetic import build as harness
    from synthetic import recipes as synth
    z, meta = load("tiny")
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, cname)
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    ids_t = torch.from_numpy(ids).cuda(); pix_t = torch.from_numpy(pix).cuda()
    mask_t = None if mask is None else torch.from_numpy(mask).cuda()
    new = 5
    out = model.generate(inputs=ids_t, images=pix_t, attention_mask=mask_t, do_sample=False, max_new_tokens=new, eos_token_id=-1).cpu()
    assert out.shape == (ids.shape[0], ids.shape[1] + new)
    w = O.to_torch_weights(synth.make_weights(cfg, 0))
    nxt = 0
    for b in range(ids.shape[0]):
        row = ids[b][mask[b].astype(bool)] if mask is not None else ids[b]
        n_img = max(1, int((row == -200).sum()))
        row_pix = pix[nxt:nxt + n_img]; nxt += n_img
        n_mark = int((row == -200).sum())
        with torch.no_grad():
            ref = O.greedy_generate(w, cfg, torch.from_numpy(row)[None], torch.from_numpy(row_pix[:max(n_mark, 1)]), new)
        assert out[b, ids.shape[1]:].tolist() == ref, (b, out[b, ids.shape[1]:].tolist(), ref)


def test_scheduler_error_isolation_and_close(cuda):
    """A request whose stopping criterion raises fails alone (model_worker turns it into error_code 1, model_worker.py:194-218);
    the others finish with the ids they produce on their own.  Closing the scheduler fails whatever is still queued."""
    import time
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    reqs = _requests(cfg, 3, seed0=700)
    alone = [model.generate(inputs=ids[None].cuda(), images=pix.cuda(), do_sample=False, max_new_tokens=8, eos_token_id=-1).cpu() for ids, pix in reqs]
    model.enable_batching(capacity=4)
    model._batcher.pause()

    class Boom:
        def __call__(self, ids, scores, **kw):
            if ids.shape[1] >= reqs[1][0].shape[0] + 3:
                raise ValueError("bad stop criterion")
            return False

    results, errors = [None] * 3, [None] * 3

    def run(i):
        ids, pix = reqs[i]
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                kw = {"stopping_criteria": [Boom()]} if i == 1 else {}
                results[i] = model.generate(inputs=ids[None].cuda(), images=pix.cuda(), do_sample=False, max_new_tokens=8, eos_token_id=-1, **kw).cpu()
        except Exception as e:  # noqa: BLE001
            errors[i] = e

    ths = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in ths: t.start()
    t0 = time.time()
    while model._batcher.queued() < 3 and time.time() - t0 < 60:
        time.sleep(0.001)
    model._batcher.resume()
    for t in ths: t.join()
    assert isinstance(errors[1], ValueError) and errors[0] is None and errors[2] is None
    assert torch.equal(results[0], alone[0]) and torch.equal(results[2], alone[2])
    # close with a request parked in the queue
    model._batcher.pause()
    parked = {}

    def late():
        try:
            model.generate(inputs=reqs[0][0][None].cuda(), images=reqs[0][1].cuda(), do_sample=False, max_new_tokens=8, eos_token_id=-1)
        except Exception as e:  # noqa: BLE001
            parked["e"] = e

    th = threading.Thread(target=late); th.start()
    t0 = time.time()
    while model._batcher.queued() < 1 and time.time() - t0 < 60:
        time.sleep(0.001)
    model.disable_batching()
    th.join(timeout=60)
    assert isinstance(parked.get("e"), RuntimeError)


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
def test_packed_prefill_equals_per_request_prefill(cuda, name):
    """generate_batch prefills its requests as ONE packed row block (lmx_prefill_batch: a single GEMM per linear and piece, attention per sequence) —
    in fp32 the ids must equal those of request-by-request prefill (lmx_prefill), for any piece size (pieces cut through sequences), with prompts of
    different lengths, a text-only request, and sampled requests (same seeds)."""
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[name]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0, weights=synth.make_weights(cfg, 0))
    prompts, images = [], []
    for i, L in enumerate((19, 27, 12, 33, 22)):
        if i == 2:
            ids = torch.from_numpy(synth.make_prompt(cfg, L, image_positions=(), seed=30 + i))       # text only
            images.append(None)
        else:
            ids = torch.from_numpy(synth.make_prompt(cfg, L, image_positions=(3 + i,), seed=30 + i))
            images.append(torch.from_numpy(synth.make_pixels(cfg, 1, seed=40 + i)).to(cuda))
        prompts.append(ids.to(cuda))
    ref = model.generate_batch(prompts, images, max_new_tokens=7, eos_token_id=-1, packed_prefill=False)
    for chunk in (0, 5, 16):
        got = model.generate_batch(prompts, images, max_new_tokens=7, eos_token_id=-1, prefill_chunk=chunk)
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), chunk
    torch.manual_seed(9); s_ref = model.generate_batch(prompts, images, max_new_tokens=6, eos_token_id=-1, do_sample=True, temperature=0.9, top_p=0.9, packed_prefill=False)
    torch.manual_seed(9); s_got = model.generate_batch(prompts, images, max_new_tokens=6, eos_token_id=-1, do_sample=True, temperature=0.9, top_p=0.9, prefill_chunk=7)
    assert all(torch.equal(a, b) for a, b in zip(s_got, s_ref))
    # and a single request through generate() (its own lmx_prefill) is the same sequence as inside the packed batch
    one = model.generate(inputs=prompts[3][None], images=images[3], do_sample=False, max_new_tokens=7, eos_token_id=-1)
    assert torch.equal(one[0], ref[3])
