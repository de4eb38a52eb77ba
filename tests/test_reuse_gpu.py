"""Turn-to-turn reuse on the GPU (llava_mi355x/reuse.py; lmx_op_hash128, lmx_seq_truncate): the LLaVA-Plus tool loop's second generate
(llava/serve/gradio_web_server_llava_plus.py:600-637) re-sends the image and the whole first exchange.  Checked: the content hash is a function of the bytes;
encode_images of pixels seen before returns the stored rows without running the tower; a second turn that extends the first one takes over its sequence,
prefills only the new rows and produces the ids a model without reuse produces (fp32 verification engine: the suffix prefill is a chunked prefill, which
test_model_gpu already holds equal to the one-shot form)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cuda, dtype, name="tiny"):
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[name]
    return cfg, harness.build_model(cfg, dtype=dtype, seed=0, weights=synth.make_weights(cfg, 0))


def test_hash128_is_a_function_of_the_bytes(cuda):
    from llava_mi355x._C import check, lib, ptr, stream_handle
    torch.manual_seed(0)
    x = torch.randn(3, 3, 336, 336, device=cuda).bfloat16()

    def h(t, items):
        out = torch.empty((items, 2), dtype=torch.int64, device=cuda)
        check(lib.lmx_op_hash128(ptr(t), t.numel() * t.element_size() // items, items, ptr(out), stream_handle()), "hash")
        return out.cpu().numpy().view(np.uint64)
    a = h(x, 3)
    assert np.array_equal(a, h(x.clone(), 3))                                   # another buffer, same bytes
    for i in range(3):
        assert np.array_equal(a[i], h(x[i].contiguous(), 1)[0])                  # batch of items == item by item
    assert len({tuple(r) for r in a.tolist()}) == 3
    y = x.clone(); y[1, 2, 300, 17] += 1.0
    b = h(y, 3)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and not np.array_equal(a[1], b[1])
    # swapping two 16-byte chunks changes the hash (position is mixed in)
    z = x[0].contiguous().view(torch.int16).flatten().clone()
    z2 = z.clone(); z2[0:8], z2[8:16] = z[8:16].clone(), z[0:8].clone()
    assert not np.array_equal(h(z, 1), h(z2, 1))
    # a length that is not a multiple of 16 bytes: the tail bytes count
    t = torch.arange(37, dtype=torch.uint8, device=cuda)
    t2 = t.clone(); t2[36] = 0
    assert not np.array_equal(h(t, 1), h(t2, 1))


def test_encode_images_cache_returns_the_stored_rows(cuda):
    from synthetic import recipes as synth
    cfg, model = _model(cuda, torch.bfloat16)
    pix = torch.from_numpy(synth.make_pixels(cfg, 3, seed=4)).to(cuda, torch.bfloat16)
    plain = model.encode_images(pix)
    model.enable_reuse(images=8, prefixes=0)
    model.profile(True)
    first = model.encode_images(pix[:2])
    n_first = model.profile_read().get("vis.embed_ln", (0.0, 0))[1]
    again = model.encode_images(torch.stack([pix[1], pix[2], pix[0]]))           # one new image between two known ones
    n_again = model.profile_read().get("vis.embed_ln", (0.0, 0))[1]
    third = model.encode_images(pix)
    n_third = model.profile_read().get("vis.embed_ln", (0.0, 0))[1]
    model.profile(False)
    assert torch.equal(first, plain[:2])
    assert torch.equal(again, torch.stack([plain[1], plain[2], plain[0]]))
    assert torch.equal(third, plain)
    assert (n_first, n_again, n_third) == (1, 1, 0)                              # tower launches: 2 new images, 1 new image, none
    st = model.reuse_stats()
    assert st["image_hits"] == 2 + 3 and st["image_misses"] == 2 + 1
    model.disable_reuse()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_second_turn_takes_over_the_first_turns_sequence(cuda, dtype):
    from llava_mi355x._C import lib
    from synthetic import recipes as synth
    cfg, model = _model(cuda, dtype)
    cfg2, fresh = _model(cuda, dtype)                                            # same weights, reuse never enabled
    ids1 = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(5,), seed=2))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=3)).to(cuda, dtype)
    P = model.tokens_per_image
    model.enable_reuse(images=4, prefixes=4, min_rows=16)
    out1 = model.generate(inputs=ids1, images=pix, do_sample=False, max_new_tokens=12, eos_token_id=-1)
    ref1 = fresh.generate(inputs=ids1, images=pix, do_sample=False, max_new_tokens=12, eos_token_id=-1)
    assert torch.equal(out1, ref1)
    assert model.reuse_stats()["prefix_entries"] == 1 and model.reuse_stats()["prefix_hits"] == 0
    # turn 2 = turn 1's prompt + its answer + the "tool output": only the new rows are prefilled
    tail = torch.from_numpy(synth.make_prompt(cfg, 23, image_positions=(), seed=9))[None].to(cuda)
    ids2 = torch.cat([out1, tail], dim=1)
    out2 = model.generate(inputs=ids2, images=pix, do_sample=False, max_new_tokens=10, eos_token_id=-1)
    ref2 = fresh.generate(inputs=ids2, images=pix, do_sample=False, max_new_tokens=10, eos_token_id=-1)
    st = model.reuse_stats()
    rows1 = ids1.shape[1] - 1 + P                                                # spliced rows of turn 1's prompt
    have = rows1 + 12 - 1                                                        # + the generated ids that own KV rows
    assert st["prefix_hits"] == 1 and st["image_hits"] == 1
    assert st["prefix_rows_reused"] == have - have % 8, (st, have)
    if dtype == torch.float32:
        assert torch.equal(out2, ref2)                                          # exact engine: chunked == one-shot
    else:
        same = (out2[0, ids2.shape[1]:] == ref2[0, ids2.shape[1]:]).float().mean().item()
        assert same >= 0.5, (out2[0, ids2.shape[1]:].tolist(), ref2[0, ids2.shape[1]:].tolist())    # bf16: a suffix prefill rounds like a chunked one
    # an unrelated conversation does not match; the finished second turn is the newest entry
    other = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(5,), seed=77))[None].to(cuda)
    pix2 = torch.from_numpy(synth.make_pixels(cfg, 1, seed=30)).to(cuda, dtype)
    o3 = model.generate(inputs=other, images=pix2, do_sample=False, max_new_tokens=4, eos_token_id=-1)
    assert torch.equal(o3, fresh.generate(inputs=other, images=pix2, do_sample=False, max_new_tokens=4, eos_token_id=-1))
    assert model.reuse_stats()["prefix_hits"] == 1 and model.reuse_stats()["prefix_entries"] == 2
    model.disable_reuse()
    assert model.reuse_stats()["prefix_entries"] == 0


def test_reused_sequence_starts_clean_after_a_device_stop_and_sampling(cuda):
    """The donor stopped by its EOS rule with steps queued ahead and drew its tokens with a sampler; the taker is greedy without a rule: nothing of the donor's
    request state (stop flag, sampling parameters, token log) may survive lmx_seq_truncate."""
    from synthetic import recipes as synth
    cfg, model = _model(cuda, torch.float32)
    _, fresh = _model(cuda, torch.float32)
    ids1 = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(5,), seed=2))[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=3)).to(cuda, torch.float32)
    free = fresh.generate(inputs=ids1, images=pix, do_sample=False, max_new_tokens=16, eos_token_id=-1)[0, ids1.shape[1]:].tolist()
    k = next((i for i in range(3, len(free)) if free[i] not in free[:i]), None)
    if k is None:
        pytest.skip("the synthetic model repeats itself")
    model.enable_reuse(images=4, prefixes=4, min_rows=16)
    a = model.generate(inputs=ids1, images=pix, do_sample=False, max_new_tokens=16, eos_token_id=free[k])      # stops at free[k] on the device, 16 steps were queued
    assert a[0, ids1.shape[1]:].tolist() == free[:k + 1]
    torch.manual_seed(0)
    ids2 = torch.cat([a, torch.from_numpy(synth.make_prompt(cfg, 9, image_positions=(), seed=5))[None].to(cuda)], dim=1)
    b = model.generate(inputs=ids2, images=pix, do_sample=True, temperature=0.9, top_p=0.8, max_new_tokens=6, eos_token_id=-1)
    assert model.reuse_stats()["prefix_hits"] == 1 and b.shape[1] == ids2.shape[1] + 6
    ids3 = torch.cat([b, torch.from_numpy(synth.make_prompt(cfg, 7, image_positions=(), seed=6))[None].to(cuda)], dim=1)
    c = model.generate(inputs=ids3, images=pix, do_sample=False, max_new_tokens=8, eos_token_id=-1)
    assert model.reuse_stats()["prefix_hits"] == 2
    assert torch.equal(c, fresh.generate(inputs=ids3, images=pix, do_sample=False, max_new_tokens=8, eos_token_id=-1))
    model.disable_reuse()
