"""Split-q decode step (csrc/decode_attn.hip: decode_kv_attn_kernel; engine option "decode_splitq") against the three-launch form it replaces.

The single-token branch of the reference (llava/model/llava_arch.py:103-112 -> HF5:models/llama/modeling_llama.py:243-281: q_proj / k_proj / v_proj, RoPE,
cache update, eager attention) as  q launch + ONE launch holding the attention workgroups and the k | v projection  must give the SAME bits as
q|k|v launch + attention launch: the GEMV rows, the RoPE, the chunk partials and the merge are the same code — only where the newest key / value come from
differs (tagged granules inside the launch instead of the row in memory).  Checked on logits after every decode step, greedy ids over positions that cross a
128-key chunk boundary, and the caches (through the ids of later steps and a re-prefill-free continuation)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decode_logits(model, ids, pix, n_steps):
    """prefill through forward(), then n_steps single-token forwards with the cache: the logits of every step + the ids picked greedily"""
    out = model.forward(input_ids=ids, images=pix, use_cache=True)
    past = out.past_key_values
    tok = out.logits[:, -1].argmax(-1, keepdim=True)
    logits, picked = [], [tok.clone()]
    for _ in range(n_steps):
        out = model.forward(input_ids=tok, past_key_values=past, use_cache=True)
        logits.append(out.logits[:, -1].clone())
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        picked.append(tok.clone())
    past.close()
    return torch.stack(logits), torch.cat(picked, 1)


@pytest.mark.parametrize("name,dtype,length,layers", [("tiny", torch.bfloat16, 20, None), ("tiny", torch.float16, 120, None), ("llava15_7b", torch.bfloat16, 512, 2),
                                                     ("llava15_7b", torch.float16, 60, 2), ("llava15_13b", torch.bfloat16, 200, 1)])
def test_splitq_decode_step_is_bit_identical(cuda, name, dtype, length, layers):
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[name]
    if layers is not None:
        cfg = synth.with_layers(cfg, layers, 1)
    model = harness.build_model(cfg, dtype=dtype, seed=0, device_rng=name != "tiny", **({} if name == "tiny" else {"max_position": 2048}))
    ids = torch.from_numpy(synth.make_prompt(cfg, length, image_positions=(5,), seed=2))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(cuda, dtype)
    try:
        model.set_option("decode_splitq", 0)
        model.profile(True); la, ia = _decode_logits(model, ids, pix, 6); names_a = set(model.profile_read()); model.profile(False)
        ga = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=40, eos_token_id=-1)
        model.set_option("decode_splitq", 1)
        model.profile(True); lb, ib = _decode_logits(model, ids, pix, 6); names_b = set(model.profile_read()); model.profile(False)
        gb = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=40, eos_token_id=-1)
    finally:
        model.set_option("decode_splitq", 1)
    assert "decode.kv_attn" in names_b and "decode.kv_attn" not in names_a and "decode.attn" in names_a, (names_a, names_b)      # both forms really ran
    assert torch.equal(la, lb), (la.float() - lb.float()).abs().max().item()
    assert torch.equal(ia, ib)
    assert torch.equal(ga, gb)


def test_splitq_many_steps_cross_chunk_boundaries_and_reuse_the_granules(cuda):
    """150 chained device-side steps from a 100-token context: the position crosses the 128- and 256-key boundaries (the live chunk count changes, the owner of
    the newest key moves to a fresh chunk), every launch reuses the sequence's granule buffer with the next tag.  Ids equal the three-launch form's."""
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=3, device_rng=True, max_position=2048)
    ids = torch.from_numpy(synth.make_prompt(cfg, 100, image_positions=(), seed=5))[None].to(cuda)
    try:
        model.set_option("decode_splitq", 0)
        ga = model.generate(inputs=ids, do_sample=False, max_new_tokens=150, eos_token_id=-1)
        model.set_option("decode_splitq", 1)
        gb = model.generate(inputs=ids, do_sample=False, max_new_tokens=150, eos_token_id=-1)
        gc = model.generate(inputs=ids, do_sample=False, max_new_tokens=150, eos_token_id=-1)
    finally:
        model.set_option("decode_splitq", 1)
    assert torch.equal(ga, gb) and torch.equal(gb, gc)


def test_splitq_steps_aside_when_many_sequences_are_alive(cuda):
    """The split-q launch parks one waiting workgroup per head; launches of different sequences may be resident together (request threads on their own
    streams), so the engine takes the form only while live sequences x heads fill at most half of the workgroups the device holds at once (occupancy of the launch x compute
    units, queried from the runtime: Model::splitq_allowed) — beyond that the three-launch form runs.  Same ids either way."""
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 1, 1)                 # 32 heads
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=1, device_rng=True, max_position=256)
    ids = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(), seed=3))[None].to(cuda)

    def run():
        model.profile(True)
        out = model.generate(inputs=ids, do_sample=False, max_new_tokens=8, eos_token_id=-1)
        names = set(model.profile_read()); model.profile(False)
        return out, names
    few, names_few = run()
    held = [LmxKVCache(model, 1) for _ in range(70)]      # 71 x 32 waiters: more than half of the device's resident workgroups at any occupancy (<= 8 per CU)
    try:
        many, names_many = run()
    finally:
        for c in held:
            c.close()
    again, names_again = run()
    assert "decode.kv_attn" in names_few and "decode.kv_attn" in names_again
    assert "decode.kv_attn" not in names_many and "decode.attn" in names_many
    assert torch.equal(few, many) and torch.equal(few, again)


def test_splitq_under_concurrent_request_threads(cuda):
    """model_worker's threading model (one generate() thread per request on its own stream, llava/serve/model_worker.py:174-185) with the split-q step: launches of
    DIFFERENT sequences are resident together, each parking its heads' mergers until its own projection has run.  8 threads x 96 tokens: every request's ids equal
    its serial run's (nothing deadlocks, no bounded wait fires, no granule of one sequence reaches another)."""
    import threading
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 4, 1)
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=2, device_rng=True, max_position=1024)
    prompts = [torch.from_numpy(synth.make_prompt(cfg, 60 + 37 * i, image_positions=(), seed=10 + i))[None].to(cuda) for i in range(8)]
    serial = [model.generate(inputs=p, do_sample=False, max_new_tokens=96, eos_token_id=-1) for p in prompts]
    out, err = [None] * len(prompts), []

    def run(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=cuda)):
                out[i] = model.generate(inputs=prompts[i], do_sample=False, max_new_tokens=96, eos_token_id=-1, run_ahead=8)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:  # noqa: BLE001
            err.append(repr(e))
    for rnd in range(3):
        ths = [threading.Thread(target=run, args=(i,)) for i in range(len(prompts))]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        assert not any(t.is_alive() for t in ths), "a request thread hangs"
        assert not err, err
        for i in range(len(prompts)):
            assert torch.equal(out[i], serial[i]), (rnd, i)


def test_splitq_wait_timeout_fails_one_sequence_once_and_falls_back(cuda):
    """a bounded in-launch wait that times out (contention, preemption, a profiler) must fail the AFFECTED sequence, once — not every later
    token read on every sequence for the life of the model.  Fault injection: the next split-q launch publishes its k | v granules under a wrong tag, so its heads'
    mergers run into the 30 ms bound, raise the sequence's own host-mapped status word and finish with a zero key.  Expected: that request raises at its next token
    read; the word is cleared; the model has switched to the three-launch form; a sequence that was alive meanwhile and every later request produce the ids of an
    undisturbed run."""
    from llava_mi355x import _C
    from synthetic import build as harness, recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 2, 1)
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=4, device_rng=True, max_position=512)
    ids = torch.from_numpy(synth.make_prompt(cfg, 50, image_positions=(), seed=7))[None].to(cuda)
    other = torch.from_numpy(synth.make_prompt(cfg, 70, image_positions=(), seed=8))[None].to(cuda)
    try:
        good = model.generate(inputs=ids, do_sample=False, max_new_tokens=12, eos_token_id=-1)
        good_other = model.generate(inputs=other, do_sample=False, max_new_tokens=12, eos_token_id=-1)
        # a bystander: prefilled and two steps in before the fault, continued after it
        out = model.forward(input_ids=other, use_cache=True)
        past = out.past_key_values
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        assert int(tok) == int(good_other[0, other.shape[1]])
        model.set_option("debug_splitq_timeout", 1)
        with pytest.raises(_C.LmxError, match="timed out"):
            model.generate(inputs=ids, do_sample=False, max_new_tokens=12, eos_token_id=-1)
        model.profile(True)
        again = model.generate(inputs=ids, do_sample=False, max_new_tokens=12, eos_token_id=-1)      # no second report, same ids as the undisturbed run
        names = set(model.profile_read()); model.profile(False)
        assert torch.equal(again, good)
        assert "decode.attn" in names and "decode.kv_attn" not in names                             # the engine stepped back to the three-launch form
        picked = [tok]
        for _ in range(3):
            o2 = model.forward(input_ids=picked[-1], past_key_values=past, use_cache=True)
            picked.append(o2.logits[:, -1].argmax(-1, keepdim=True))
        assert torch.cat(picked, 1)[0].tolist() == good_other[0, other.shape[1]: other.shape[1] + 4].tolist()
        past.close()
    finally:
        model.set_option("debug_splitq_timeout", 0)
        model.set_option("decode_splitq", 1)
