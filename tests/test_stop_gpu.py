"""Device-side stop rule (include/llava_mi355x.h: lmx_seq_set_stop; csrc/sampling.hip: pick_advance_batch_kernel).

The reference tests every new token on the HOST: `eos_token_id` inside HF generate() and KeywordsStoppingCriteria's id rule
`(output_ids[0, -len(kw):] == kw).all()` (llava/mm_utils.py:94-107, built from the worker's "stop" string at model_worker.py:160-165) — one D2H copy
per token.  Here both id rules travel with the sequence; the pick kernel applies them to the token it just appended, and a sequence whose rule
fired is not advanced by later picks.  Checked: steps chained on the device past the stop produce nothing (token log, position of the device state);
a decode batch reports -1 for a stopped member while the others go on; generate() / generate_batch() return exactly the ids the host-only rule
returned before (the criterion objects are still evaluated on the host — same answers, no discarded ids)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cuda, name="tiny", dtype=torch.bfloat16):
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[name]
    return cfg, harness.build_model(cfg, dtype=dtype, seed=0, weights=synth.make_weights(cfg, 0))


def _request(cfg, cuda, dtype, length=24, seed=2):
    from synthetic import recipes as synth
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(5,), seed=seed))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=seed + 1)).to(cuda, dtype)
    return ids, pix


def _read(model, seq, cap=256):
    from llava_mi355x._C import check, lib, stream_handle
    host = (ctypes.c_int64 * cap)(); n = ctypes.c_int32(0)
    check(lib.lmx_seq_read_tokens(seq, host, cap, ctypes.byref(n), stream_handle()), "read_tokens")
    return [int(host[i]) for i in range(n.value)]


def _stopped(seq):
    from llava_mi355x._C import check, lib, stream_handle
    f = ctypes.c_int32(0)
    check(lib.lmx_seq_stopped(seq, ctypes.byref(f), stream_handle()), "stopped")
    return bool(f.value)


def _free_run(model, ids, pix, n):
    return model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=n, eos_token_id=-1)[0, ids.shape[1]:].tolist()


def _first_new(tokens, start=3):
    """index >= start of a token that does not occur earlier in the run (so that a rule on it fires exactly there)"""
    for i in range(start, len(tokens)):
        if tokens[i] not in tokens[:i]:
            return i
    pytest.skip("the synthetic model repeats itself: no usable stop token")


def test_chained_steps_produce_nothing_past_an_eos(cuda):
    from llava_mi355x._C import check, lib, stream_handle
    cfg, model = _model(cuda)
    ids, pix = _request(cfg, cuda, torch.bfloat16)
    free = _free_run(model, ids, pix, 24)
    k = _first_new(free)
    cache = model._prefill_request(ids, pix, None, None, stop=([free[k]], []))
    seq = cache.seqs[0]
    try:
        assert not _stopped(seq)
        check(lib.lmx_decode(model._h, seq, -1, 23, None, 1, stream_handle()), "decode")      # 23 steps chained on the device, no host in between
        got = _read(model, seq)
        assert got == free[:k + 1], (got, free)                    # ends WITH the EOS id, nothing after it
        assert _stopped(seq)
        # more steps: still nothing
        check(lib.lmx_decode(model._h, seq, -1, 5, None, 1, stream_handle()), "decode")
        assert _read(model, seq) == free[:k + 1]
    finally:
        cache.close()
    # a recycled sequence starts without a rule
    cache = model._prefill_request(ids, pix, None, None)
    try:
        check(lib.lmx_decode(model._h, cache.seqs[0], -1, 23, None, 1, stream_handle()), "decode")
        assert _read(model, cache.seqs[0]) == free
        assert not _stopped(cache.seqs[0])
    finally:
        cache.close()


def test_keyword_ids_rule_and_first_pick(cuda):
    from llava_mi355x._C import check, lib, stream_handle
    cfg, model = _model(cuda, "tiny_gqa")
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=30, seed=4)
    free = _free_run(model, ids, pix, 24)
    k = _first_new(free, start=4)
    # two keywords: one that never occurs, one = the three ids ending at position k
    kw = [[free[k], free[k]], free[k - 2:k + 1]]
    cache = model._prefill_request(ids, pix, None, None, stop=([], kw))
    try:
        check(lib.lmx_decode(model._h, cache.seqs[0], -1, 23, None, 1, stream_handle()), "decode")
        assert _read(model, cache.seqs[0]) == free[:k + 1]
        assert _stopped(cache.seqs[0])
    finally:
        cache.close()
    # the prefill's own pick (token 1) is tested too: an EOS equal to it stops the sequence before any decode step
    cache = model._prefill_request(ids, pix, None, None, stop=([free[0]], []))
    try:
        check(lib.lmx_decode(model._h, cache.seqs[0], -1, 8, None, 1, stream_handle()), "decode")
        assert _read(model, cache.seqs[0]) == free[:1]
    finally:
        cache.close()


def test_decode_batch_reports_minus_one_for_a_stopped_member(cuda):
    from llava_mi355x.batching import DecodeBatch
    cfg, model = _model(cuda)
    reqs = [_request(cfg, cuda, torch.bfloat16, length=20 + 4 * i, seed=10 + i) for i in range(3)]
    free = [_free_run(model, i, p, 16) for i, p in reqs]
    k = _first_new(free[1])
    caches = [model._prefill_request(i, p, None, None, stop=(([free[1][k]], []) if j == 1 else None)) for j, (i, p) in enumerate(reqs)]
    batch = DecodeBatch(model, 4)
    try:
        steps = batch.step([c.seqs[0] for c in caches], None, 15, True)          # ids[step][member], token 1 came from the prefill
        for j in (0, 2):
            assert [s[j] for s in steps] == free[j][1:]
        mid = [s[1] for s in steps]
        assert mid[:k] == free[1][1:k + 1] and all(v == -1 for v in mid[k:]), (mid, free[1], k)
        assert _read(model, caches[1].seqs[0]) == free[1][:k + 1]
    finally:
        batch.close()
        for c in caches:
            c.close()


def test_lone_batch_member_reports_minus_one_after_its_stop(cuda):
    """a decode batch with ONE member takes the single-sequence step; its chained steps must report ids exactly like a member of a larger batch — the picked
    ids up to and including the stop token, then -1 — and leave the host's position mirror on the device's (the next call continues from the right slot)."""
    from llava_mi355x._C import lib
    from llava_mi355x.batching import DecodeBatch
    cfg, model = _model(cuda)
    ids, pix = _request(cfg, cuda, torch.bfloat16, length=22, seed=31)
    free = _free_run(model, ids, pix, 16)
    k = _first_new(free)
    cache = model._prefill_request(ids, pix, None, None, stop=([free[k]], []))
    batch = DecodeBatch(model, 4)
    try:
        len0 = lib.lmx_seq_length(cache.seqs[0])
        steps = batch.step([cache.seqs[0]], None, 15, True)                     # [step][member]
        got = [s[0] for s in steps]
        assert got[:k] == free[1:k + 1] and all(v == -1 for v in got[k:]), (got, free, k)
        assert _read(model, cache.seqs[0]) == free[:k + 1]
        assert _stopped(cache.seqs[0])
        assert lib.lmx_seq_length(cache.seqs[0]) == len0 + k                      # k tokens were appended by this call; the steps behind the stop moved nothing
    finally:
        batch.close()
        cache.close()


class _Keywords:
    """KeywordsStoppingCriteria's id rule (llava/mm_utils.py:94-107) without a tokenizer"""
    def __init__(self, keyword_ids):
        self.keyword_ids = [torch.tensor(k) for k in keyword_ids]
        self.calls = 0

    def __call__(self, output_ids, scores, **kw):
        self.calls += 1
        return any(output_ids.shape[1] >= k.shape[0] and bool((output_ids[0, -k.shape[0]:].cpu() == k).all()) for k in self.keyword_ids)


@pytest.mark.parametrize("run_ahead", [1, 16])
def test_generate_returns_the_same_ids_as_the_host_only_rule(cuda, run_ahead):
    cfg, model = _model(cuda)
    ids, pix = _request(cfg, cuda, torch.bfloat16)
    free = _free_run(model, ids, pix, 24)
    k = _first_new(free)
    L = ids.shape[1]
    out = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=free[k], run_ahead=run_ahead)
    assert out[0, L:].tolist() == free[:k + 1]
    crit = _Keywords([free[k - 1:k + 1]])
    out = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1, stopping_criteria=[crit], run_ahead=run_ahead)
    assert out[0, L:].tolist() == free[:k + 1]
    assert crit.calls == k + 1                                     # evaluated once per produced token — and no token was produced past the stop
    # a rule the device cannot hold (a 9-id keyword) is the host's alone: same answer
    long_kw = _Keywords([[7] * 9, free[k - 1:k + 1]])
    assert model._stop_spec(set(), [long_kw]) is None
    out = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=24, eos_token_id=-1, stopping_criteria=[long_kw], run_ahead=run_ahead)
    assert out[0, L:].tolist() == free[:k + 1]


def test_generate_batch_and_scheduler_with_a_member_that_stops(cuda):
    cfg, model = _model(cuda)
    reqs = [_request(cfg, cuda, torch.bfloat16, length=20 + 4 * i, seed=10 + i) for i in range(3)]
    free = [_free_run(model, i, p, 16) for i, p in reqs]
    k = _first_new(free[1])
    eos = free[1][k]
    want = [f[:f.index(eos) + 1] if eos in f else f for f in free]
    outs = model.generate_batch([i[0] for i, _ in reqs], [p for _, p in reqs], max_new_tokens=16, eos_token_id=eos, run_ahead=8)
    for (i, _), o, w in zip(reqs, outs, want):
        assert o[i.shape[1]:].tolist() == w
    # continuous batching: the scheduler's two-deep pipeline enqueues one more step for a member before the host has seen its EOS
    import threading
    model.enable_batching(capacity=4)
    try:
        got = [None] * 3

        def run(j):
            got[j] = model.generate(inputs=reqs[j][0], images=reqs[j][1], do_sample=False, max_new_tokens=16, eos_token_id=eos)[0, reqs[j][0].shape[1]:].tolist()
        ths = [threading.Thread(target=run, args=(j,)) for j in range(3)]
        for t in ths: t.start()
        for t in ths: t.join(timeout=120)
        assert got == want
    finally:
        model.disable_batching()


def test_host_length_follows_the_device_after_a_stop_with_run_ahead(cuda):
    """steps queued ahead of a device-side stop advanced only the HOST mirror of the position.  After the host has observed the stop
    (lmx_seq_read_tokens / lmx_seq_stopped) lmx_seq_length equals prompt + generated — so a C caller that continues the sequence (forward with
    past_key_values, another decode) appends at the stop position — and lmx_seq_reset re-arms a stopped sequence."""
    from llava_mi355x._C import check, lib, stream_handle
    cfg, model = _model(cuda)
    ids, pix = _request(cfg, cuda, torch.bfloat16)
    free = _free_run(model, ids, pix, 24)
    k = _first_new(free)
    cache = model._prefill_request(ids, pix, None, None, stop=([free[k]], []))
    seq = cache.seqs[0]
    try:
        n_prompt = lib.lmx_seq_length(seq)
        check(lib.lmx_decode(model._h, seq, -1, 23, None, 1, stream_handle()), "decode")      # run-ahead: 23 steps queued, the rule fires after k
        got = _read(model, seq)
        assert got == free[:k + 1]
        # the prefill produced token 0; decode step j feeds token j-1 at position n_prompt + j - 1: k steps ran before the stop froze the position
        assert lib.lmx_seq_length(seq) == n_prompt + k, (lib.lmx_seq_length(seq), n_prompt, k)
        assert _stopped(seq) and lib.lmx_seq_length(seq) == n_prompt + k
        # more queued steps do not drift the mirror once the stop has been observed
        check(lib.lmx_decode(model._h, seq, -1, 4, None, 1, stream_handle()), "decode")
        assert _stopped(seq) and lib.lmx_seq_length(seq) == n_prompt + k
    finally:
        cache.close()
