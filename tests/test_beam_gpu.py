"""Beam search on the device path (`generate(num_beams > 1)`; llava_mi355x/beam.py, lmx_op_beam_topk, lmx_seq_copy) — llava/eval/run_llava.py:121 and
model_vqa_loader.py:104 pass `num_beams` through to the reference's generate.

* against tests/golden/beam.npz: ids the REFERENCE model's own generate(num_beams=k) returned (oracle/make_golden_beam.py, no-EOS cases) — fp32 engine,
  ids must be identical (in 4 of the 5 cases beam search returns something else than greedy decoding);
* against oracle/beam_oracle.py (pinned to the reference by tests/test_beam_oracle_vs_reference.py) with an EOS id that really closes hypotheses early
  (transformers-4.31 length rules), length_penalty 1.0 and 2.0, early_stopping False / True;
* the top-K kernel alone against torch.log_softmax + topk; a cache copy must reproduce its source's continuation."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _case(meta):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[meta["config"]]
    ids = torch.from_numpy(synth.make_prompt(cfg, meta["prompt_len"], image_positions=(meta["image_pos"],), seed=meta["seed_ids"]))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=meta["seed_pix"]))
    return cfg, ids, pix


def test_beam_ids_equal_the_reference_goldens(cuda):
    from synthetic import build as harness, recipes as synth
    z = np.load(os.path.join(HERE, "golden", "beam.npz"))
    meta = json.loads(bytes(z["meta"]).decode())["cases"]
    differs = 0
    for i, m in enumerate(meta):
        cfg, ids, pix = _case(m)
        model = harness.build_model(cfg, dtype=torch.float32, seed=0, weights=synth.make_weights(cfg, 0))
        out = model.generate(inputs=ids.to(cuda), images=pix.to(cuda), do_sample=False, num_beams=m["num_beams"], max_new_tokens=m["max_new_tokens"], eos_token_id=-1)
        assert out[0, : ids.shape[1]].cpu().tolist() == ids[0].tolist()                      # the prompt (markers included) is echoed
        assert out[0, ids.shape[1]:].cpu().tolist() == z[f"case{i}.beam"].tolist(), (i, m)
        differs += int(z[f"case{i}.beam"].tolist() != z[f"case{i}.greedy"].tolist())
    assert differs >= 2


@pytest.mark.parametrize("lp,early", [(1.0, False), (2.0, False), (1.0, True)])
def test_beam_with_eos_matches_oracle(cuda, lp, early):
    from oracle import beam_oracle, llava_oracle as O
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    model = harness.build_model(cfg, dtype=torch.float32, seed=0, weights=wnp)
    ids = torch.from_numpy(synth.make_prompt(cfg, 20, image_positions=(5,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=3))
    free = beam_oracle.beam_search(w, cfg, ids, pix, 3, 8)
    closed = 0
    for eos in (free[2], free[4]):                       # ids the unconstrained search emits at steps 3 and 5: as EOS they end hypotheses early
        want = beam_oracle.beam_search(w, cfg, ids, pix, 3, 8, eos_ids=[eos], length_penalty=lp, early_stopping=early)
        got = model.generate(inputs=ids.to(cuda), images=pix.to(cuda), do_sample=False, num_beams=3, max_new_tokens=8, eos_token_id=int(eos),
                             length_penalty=lp, early_stopping=early)[0, ids.shape[1]:].cpu().tolist()
        assert got == want, (eos, got, want)
        closed += int(len(want) < 8 or eos in want)
    assert closed >= 1


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_beam_topk_kernel(cuda, dt):
    from llava_mi355x import _C
    torch.manual_seed(0)
    B, V, Vp, K = 4, 1000, 1008, 8
    logits = (torch.randn(B, Vp) * 3).to(dt)
    logits[1, 17] = logits[1, 500]                        # a tie: lower id first
    beam = torch.tensor([0.0, -1.5, -0.25, -7.0])
    sc = torch.empty((B, K), dtype=torch.float32, device=cuda); ix = torch.empty((B, K), dtype=torch.int32, device=cuda)
    ld = logits.to(cuda)
    _C.check(_C.lib.lmx_op_beam_topk(_C.torch_dtype_code(dt), _C.ptr(ld), Vp, V, B, _C.ptr(beam.to(cuda)), K, _C.ptr(sc), _C.ptr(ix), _C.stream_handle()))
    ref = torch.log_softmax(logits[:, :V].float(), dim=-1) + beam[:, None]
    for b in range(B):
        order = sorted(range(V), key=lambda i: (-float(ref[b, i]), i))[:K]
        assert ix[b].cpu().tolist() == order
        assert (sc[b].cpu() - ref[b, order]).abs().max().item() <= 1e-5


def test_seq_copy_continues_like_its_source(cuda):
    from llava_mi355x import _C
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny_gqa"]
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, weights=synth.make_weights(cfg, 0))
    ids = torch.from_numpy(synth.make_prompt(cfg, 22, image_positions=(4,), seed=5))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=6)).to(cuda, torch.bfloat16)
    a, _ = model._prefill_request(ids, pix, None, None, 0, return_logits=True)
    b = LmxKVCache(model, 1)
    _C.check(_C.lib.lmx_seq_copy(b.seqs[0], a.seqs[0], _C.stream_handle()))
    assert _C.lib.lmx_seq_length(b.seqs[0]) == _C.lib.lmx_seq_length(a.seqs[0])
    V = model._vocab_cap
    la = torch.empty((1, V), dtype=torch.bfloat16, device=cuda); lb = torch.empty_like(la)
    for tok in (7, 19, 3):
        _C.check(_C.lib.lmx_decode(model._h, a.seqs[0], tok, 1, _C.ptr(la), 0, _C.stream_handle()))
        _C.check(_C.lib.lmx_decode(model._h, b.seqs[0], tok, 1, _C.ptr(lb), 0, _C.stream_handle()))
        assert torch.equal(la, lb)
    a.close(); b.close()


class _Keywords:
    """KeywordsStoppingCriteria's id rule over a batch of rows (llava/mm_utils.py:94-114: every row must end with a keyword's ids), no tokenizer"""
    def __init__(self, keyword_ids):
        self.keyword_ids = [torch.tensor(k) for k in keyword_ids]

    def __call__(self, output_ids, scores, **kw):
        return all(any(bool((row[-k.shape[0]:].cpu() == k).all()) for k in self.keyword_ids) for row in output_ids)


def test_beam_with_stopping_criteria_and_image_batches(cuda):
    """VERDICT r2 missing #5: generate(num_beams > 1) used to raise for stopping_criteria and for a batch of rows with images.  Criteria follow
    transformers 4.31's beam loop (oracle/beam_oracle.py); a batch is searched row by row, each row with its own images (llava_arch.py:150-159)."""
    from oracle import beam_oracle, llava_oracle as O
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    model = harness.build_model(cfg, dtype=torch.float32, seed=0, weights=wnp)
    ids = torch.from_numpy(synth.make_prompt(cfg, 20, image_positions=(5,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=3))
    free = beam_oracle.beam_search(w, cfg, ids, pix, 2, 8)
    # a criterion that fires when EVERY continuing beam ends with the id the best hypothesis has at step 3 (with 2 beams that happens or not: both
    # answers must agree with the oracle), and one that fires on any row (a custom criterion)
    class _AnyRow(_Keywords):
        def __call__(self, output_ids, scores, **kw):
            return any(any(bool((row[-k.shape[0]:].cpu() == k).all()) for k in self.keyword_ids) for row in output_ids)
    stopped = 0
    for crit in (_Keywords([[free[2]]]), _AnyRow([[free[2]]]), _AnyRow([free[3:5]])):
        want = beam_oracle.beam_search(w, cfg, ids, pix, 2, 8, stopping_criteria=[crit])
        got = model.generate(inputs=ids.to(cuda), images=pix.to(cuda), do_sample=False, num_beams=2, max_new_tokens=8, eos_token_id=-1,
                             stopping_criteria=[crit])[0, ids.shape[1]:].cpu().tolist()
        assert got == want, (got, want)
        stopped += int(len(want) < 8)
    assert stopped >= 1
    # two rows, each with its own image, searched with beams: row results equal the single-row calls
    ids2 = torch.from_numpy(synth.make_prompt(cfg, 20, image_positions=(7,), seed=5))[None]
    pix2 = torch.from_numpy(synth.make_pixels(cfg, 1, seed=6))
    one = [model.generate(inputs=i.to(cuda), images=p.to(cuda), do_sample=False, num_beams=3, max_new_tokens=6, eos_token_id=-1)[0, 20:].cpu().tolist()
           for i, p in ((ids, pix), (ids2, pix2))]
    both = model.generate(inputs=torch.cat([ids, ids2]).to(cuda), images=torch.cat([pix, pix2]).to(cuda), do_sample=False, num_beams=3, max_new_tokens=6, eos_token_id=-1)
    assert [both[0, 20:].cpu().tolist(), both[1, 20:].cpu().tolist()] == one


def _beam_sample_topk(logits, keep, beam_scores, T, seed, counter0, K):
    from llava_mi355x import _C
    B, V = logits.shape
    dev = logits.device
    keys = torch.empty((B, K), dtype=torch.float32, device=dev); sc = torch.empty((B, K), dtype=torch.float32, device=dev)
    ix = torch.empty((B, K), dtype=torch.int32, device=dev)
    bs = None if beam_scores is None else beam_scores.to(dev, torch.float32)
    _C.check(_C.lib.lmx_op_beam_sample_topk(_C.torch_dtype_code(logits.dtype), _C.ptr(logits), logits.stride(0), V, B, _C.ptr(keep) if keep is not None else None,
                                            _C.ptr(bs) if bs is not None else None, float(T), int(seed), int(counter0), K, _C.ptr(keys), _C.ptr(sc), _C.ptr(ix),
                                            _C.stream_handle()), "lmx_op_beam_sample_topk")
    torch.cuda.synchronize()
    return keys.cpu(), sc.cpu(), ix.cpu()


def test_beam_sample_topk_kernel_scores_order_and_distribution(cuda):
    """lmx_op_beam_sample_topk: (1) the scores it reports are (log_softmax + beam_score) / T of the ids it reports; (2) keys descend, no id twice, only ids the
    keep mask allows; (3) the same (seed, counter) repeats bit for bit, another counter draws differently; (4) DISTRIBUTION: over 6000 counters the first draw over
    the whole beams x V block (largest key) follows softmax of the warped scores, and the second draw follows the Plackett-Luce conditional (softmax over the
    rest) — the law of torch.multinomial(replacement=False) that GenerationMixin.beam_sample draws from."""
    B, V, K, T = 2, 24, 4, 0.7
    g = torch.Generator().manual_seed(11)
    logits = (torch.randn(B, V, generator=g) * 1.5).to(cuda)
    bs = torch.tensor([0.0, -0.8])
    keep = torch.ones((B, V), dtype=torch.uint8); keep[0, 3] = 0; keep[1, 5:9] = 0
    keep_d = keep.to(cuda)
    s = ((torch.log_softmax(logits.double().cpu(), -1) + bs[:, None].double()) / T)
    s_masked = s.masked_fill(keep == 0, -float("inf"))
    keys, sc, ix = _beam_sample_topk(logits, keep_d, bs, T, 1234, 0, K)
    for b in range(B):
        ids = ix[b].tolist()
        assert len(set(ids)) == K and all(keep[b, i] for i in ids)
        assert all(keys[b, k] >= keys[b, k + 1] for k in range(K - 1))
        assert torch.allclose(sc[b].double(), s[b, ix[b].long()], atol=2e-5, rtol=1e-5)
    again = _beam_sample_topk(logits, keep_d, bs, T, 1234, 0, K)
    assert all(torch.equal(a, b) for a, b in zip((keys, sc, ix), again))
    other = _beam_sample_topk(logits, keep_d, bs, T, 1234, B * V, K)
    assert not torch.equal(other[2], ix)
    # distribution of the first two draws over the joint block
    p = torch.softmax(s_masked.reshape(-1), 0)                      # [B * V]
    N = 6000
    first = torch.zeros(B * V, dtype=torch.float64); pair_ok = 0.0
    second_given = {}
    for n in range(N):
        ky, _, ids = _beam_sample_topk(logits, keep_d, bs, T, 99, n * B * V, K)
        flat = [(float(ky[b, k]), b * V + int(ids[b, k])) for b in range(B) for k in range(K)]
        flat.sort(key=lambda c: -c[0])
        first[flat[0][1]] += 1
        second_given.setdefault(flat[0][1], torch.zeros(B * V, dtype=torch.float64))[flat[1][1]] += 1
    emp = first / N
    assert float((emp - p).abs().max()) < 4.0 * float((p * (1 - p) / N).sqrt().max()) + 2e-3, (emp, p)
    assert float(emp[p == 0].sum()) == 0.0
    top = int(p.argmax())                                            # conditional law of the second draw given the most frequent first draw
    cnt = second_given[top]; n_top = float(cnt.sum())
    q = p.clone(); q[top] = 0; q = q / q.sum()
    assert n_top > 500 and float((cnt / n_top - q).abs().max()) < 4.0 * float((q * (1 - q) / n_top).sqrt().max()) + 5e-3


def test_beam_sample_generate(cuda):
    """generate(num_beams > 1, do_sample=True) = GenerationMixin.beam_sample: reproducible under torch.manual_seed, different under another seed; ONE step at
    a very low temperature is plain beam search's step (the noise cannot reorder scores divided by 1e-3; over several steps the comparison is meaningless:
    beam_sample feeds the temperature-scaled score back, so it grows by 1 / T per step — in transformers as well); top-p / top-k go through the survivor masks."""
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    model = harness.build_model(cfg, dtype=torch.float32, seed=0, weights=synth.make_weights(cfg, 0))
    ids = torch.from_numpy(synth.make_prompt(cfg, 10, image_positions=(4,), seed=5))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=6)).to(cuda)
    kw = dict(inputs=ids, images=pix, num_beams=3, max_new_tokens=8, eos_token_id=-1)
    beam = model.generate(do_sample=False, **kw)
    torch.manual_seed(0)
    kw1 = dict(kw, max_new_tokens=1)
    assert torch.equal(model.generate(do_sample=True, temperature=1e-3, **kw1), model.generate(do_sample=False, **kw1))
    outs = []
    for seed in (1, 1, 2, 3, 4):
        torch.manual_seed(seed)
        outs.append(model.generate(do_sample=True, temperature=1.5, top_p=0.95, top_k=40, **kw))
        assert outs[-1].shape == beam.shape and torch.equal(outs[-1][:, : ids.shape[1]], ids)
        assert int(outs[-1][0, ids.shape[1]:].min()) >= 0 and int(outs[-1][0, ids.shape[1]:].max()) < cfg.vocab_size
    assert torch.equal(outs[0], outs[1])
    assert len({tuple(o[0].tolist()) for o in outs}) >= 3            # hot sampling over a random-weight model: seeds disagree
    torch.manual_seed(7)
    nucleus = model.generate(do_sample=True, temperature=1.0, top_p=1e-6, **kw)     # min_tokens_to_keep = 2 per beam row: still 2 * num_beams candidates per step
    assert nucleus.shape == beam.shape
