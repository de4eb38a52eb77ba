"""Parity of the decode BATCH's attention launch — the kernel every serving number at >= 16 sequences runs on (VERDICT r5 "What's weak" 1).

`launch_decode_fused` with a per-sequence table (csrc/attention.hip) sends 16-bit models with head_dim 128 to `decode_attn_wave_kernel<T, NWV, HPW>`
(csrc/attention_batch.h; round 5 launched 8 waves x 2 heads per workgroup from n_heads x n_seq >= 512 on and 8 x 1 below, round 6 measured 4 x 1 the best form at
every batch size and launches only that one).  The reference has no
batching (llava/serve/model_worker.py:174-185: one generate() thread per request); the contract per sequence is the single-token branch of
llava/model/llava_arch.py:103-112 -> HF5:models/llama/modeling_llama.py:191-214 (eager attention, softmax in fp32) and :243-281 (RoPE on q and the new key,
KV append).  Op level: `lmx_op_decode_attn_batch` against a float64 statement of that arithmetic for every sequence of the launch, with per-sequence
positions mixed over chunk / piece boundaries and the cache's last slot, caches holding garbage (stale finite values, or NaN / Inf) past `pos`, multi-head
and grouped-query geometries.  Engine level: 16 requests decoding together on a one-layer model at the real 7B widths against the oracle."""
import math

import pytest
import torch

from test_ops_gpu import DT, TOL, _attn_ref, _fill_cache, _rel_err, _rope_ref, _rope_table

pytestmark = pytest.mark.gpu

POSITIONS = (0, 63, 64, 127, 1087, 1215, 2047, 1, 128, 640)


def _build(cuda, T, n_seq, nh, nkv, D, s_max, garbage, seed):
    g = torch.Generator(device=cuda).manual_seed(seed)
    pos = [POSITIONS[(i + seed) % len(POSITIONS)] for i in range(n_seq)]
    qkv = torch.randn(n_seq, (nh + 2 * nkv) * D, device=cuda, generator=g).to(T)
    seqs = []
    for z in range(n_seq):
        p = pos[z]
        k_past = torch.randn(p, nkv, D, device=cuda, generator=g).to(T); v_past = torch.randn(p, nkv, D, device=cuda, generator=g).to(T)
        kc, vt = _fill_cache(k_past, v_past, s_max, T, cuda)
        if garbage == "stale":          # what lmx_seq_truncate / a pooled sequence leaves behind: finite values of an earlier request
            kc[:, p:] = (torch.randn(nkv, s_max - p, D, device=cuda, generator=g) * 50).to(T)
            vt[:, :, p:] = (torch.randn(nkv, D, s_max - p, device=cuda, generator=g) * 50).to(T)
        elif garbage == "nonfinite":    # nothing past `pos` may reach the result, not even through 0 x Inf
            kc[:, p:] = float("nan")
            vt[:, :, p:] = float("inf")
            vt[:, :, p::3] = float("nan")
        seqs.append((k_past, v_past, kc, vt, torch.tensor([p], dtype=torch.int32, device=cuda)))
    return pos, qkv, seqs


def _check(cuda, dt, n_seq, nh, nkv, garbage, seed=0):
    from llava_mi355x import ops
    T, D, s_max = DT[dt], 128, 2048
    table = _rope_table(s_max, D).to(cuda)
    pos, qkv, seqs = _build(cuda, T, n_seq, nh, nkv, D, s_max, garbage, seed)
    before = [(kc.clone(), vt.clone()) for _, _, kc, vt, _ in seqs]
    src = qkv.clone()
    out = ops.decode_attn_batch(qkv, [s[2] for s in seqs], [s[3] for s in seqs], [s[4] for s in seqs], table, nh, nkv, D, n_split=max(pos) // 128 + 1)
    torch.cuda.synchronize()
    assert torch.equal(qkv, src)                                             # the q | k | v rows are inputs only
    tb = table.to(T).double()                                                # HF: cos / sin in the model dtype
    worst = 0.0
    for z, (k_past, v_past, kc, vt, _) in enumerate(seqs):
        p = pos[z]
        p1 = torch.tensor([p], device=cuda)
        q_r = _rope_ref(src[z, : nh * D].double().view(1, nh, D), p1, tb, D).to(T)
        k_r = _rope_ref(src[z, nh * D:(nh + nkv) * D].double().view(1, nkv, D), p1, tb, D).to(T)
        v_n = src[z, (nh + nkv) * D:].view(1, nkv, D)
        k_all = torch.cat([k_past, k_r], 0).double(); v_all = torch.cat([v_past, v_n], 0).double()
        ref = _attn_ref(q_r.double(), k_all, v_all, False, 0).reshape(nh * D)
        assert torch.isfinite(out[z].float()).all(), f"sequence {z} (pos {p}): non-finite output"
        err = _rel_err(out[z], ref)
        worst = max(worst, err)
        assert err < TOL[dt], f"sequence {z} (pos {p}): {err:.3e}"
        # the append: rotated k row and the exact v column at `pos`; every other byte of both caches as it was (prefix AND the garbage behind)
        assert _rel_err(kc[:, p], k_r[0]) < TOL[dt]
        assert torch.equal(vt[:, :, p], v_n[0])
        kb, vb = before[z]
        keep = torch.ones(s_max, dtype=torch.bool, device=cuda); keep[p] = False
        assert torch.equal(kc[:, keep].view(torch.int16), kb[:, keep].view(torch.int16))
        assert torch.equal(vt[:, :, keep].view(torch.int16), vb[:, :, keep].view(torch.int16))
    return out, worst


# (sequences, heads, kv heads): the serving sizes (16 / 32 sequences x 32 / 40 heads), smaller batches, grouped-query geometries (4 and 8 query heads per kv head: only
# the group's first head appends the new key / value).  Every launch mixes the positions 0 (empty cache: the new key alone), 63 / 64 (last key of a 64-key piece / first of the
# next), 127 / 128, 640, 1087 / 1215 (first / last decode step of the bench) and 2047 (the cache's last slot).
GEOMETRIES = [(16, 32, 32), (32, 32, 32), (16, 40, 40), (32, 40, 40), (8, 32, 32), (12, 40, 40), (16, 32, 8), (4, 32, 8), (32, 32, 4)]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("garbage", ["zero", "stale", "nonfinite"])
@pytest.mark.parametrize("n_seq,nh,nkv", GEOMETRIES)
def test_decode_attn_batch_vs_fp64(cuda, dt, garbage, n_seq, nh, nkv):
    _check(cuda, dt, n_seq, nh, nkv, garbage, seed=n_seq + nh)


@pytest.mark.parametrize("n_seq,nh,nkv", [(16, 32, 32), (8, 32, 32), (16, 32, 8)])
def test_decode_attn_batch_is_deterministic_and_order_free(cuda, n_seq, nh, nkv):
    """Run to run the launch is bit-identical (fixed merge order), and a sequence's row does not depend on which strangers share the launch or on its slot:
    the same sequences in reversed order give the same bits, and each one alone in a launch agrees with its row of the full launch."""
    from llava_mi355x import ops
    T, D, s_max = torch.bfloat16, 128, 2048
    table = _rope_table(s_max, D).to(cuda)
    pos, qkv, seqs = _build(cuda, T, n_seq, nh, nkv, D, s_max, "stale", 3)
    snap = [(s[2].clone(), s[3].clone()) for s in seqs]

    def run(order):
        for (kc0, vt0), s in zip(snap, seqs):
            s[2].copy_(kc0); s[3].copy_(vt0)
        o = ops.decode_attn_batch(qkv[order].contiguous(), [seqs[i][2] for i in order], [seqs[i][3] for i in order], [seqs[i][4] for i in order], table, nh, nkv, D)
        torch.cuda.synchronize()
        return o

    fwd = list(range(n_seq))
    a = run(fwd); b = run(fwd)
    assert torch.equal(a, b)
    rev = fwd[::-1]
    c = run(rev)
    assert torch.equal(c, a[rev])
    for z in (0, n_seq // 2, n_seq - 1):
        lone = run([z])
        assert _rel_err(lone[0], a[z]) < TOL["bf16"]


def test_decode_attn_batch_matches_single_request_launch(cuda):
    """The batch kernel (online softmax over 64-key pieces) and the single request's launch (decode_attn_step_kernel: 128-key partials merged in chunk order) sum in
    different orders, so they agree to rounding, not bit for bit; both are held to the float64 statement above — here to each other."""
    from llava_mi355x import ops
    T, D, s_max, nh, nkv, n_seq = torch.bfloat16, 128, 2048, 32, 32, 16
    table = _rope_table(s_max, D).to(cuda)
    pos, qkv, seqs = _build(cuda, T, n_seq, nh, nkv, D, s_max, "zero", 5)
    snap = [(s[2].clone(), s[3].clone()) for s in seqs]
    out = ops.decode_attn_batch(qkv, [s[2] for s in seqs], [s[3] for s in seqs], [s[4] for s in seqs], table, nh, nkv, D)
    for z in range(n_seq):
        kc, vt = snap[z]
        one = ops.decode_attn_step(qkv[z].clone(), kc, vt, table, pos[z], nh, nkv, D)
        assert _rel_err(out[z], one) < TOL["bf16"], (z, pos[z])
        assert torch.equal(vt, seqs[z][3])                       # same V^T append
        assert _rel_err(kc[:, pos[z]], seqs[z][2][:, pos[z]]) < 1e-6


@pytest.mark.parametrize("n_req", [16, 32])
def test_sixteen_requests_decode_together_vs_oracle(cuda, n_req):
    """Engine level: n_req requests (own image, own prompt length) prefilled together and decoded TOGETHER on a one-layer model at the real 7B widths
    (32 heads x 128: every batched step's attention is decode_attn_wave_kernel) against the CPU oracle: every id a request produced is, for the
    oracle fed that request's own prefix, within 3e-2 of max|logit| of the oracle's best logit (bf16 may pick another near-tie; the oracle is fp32), and the
    batched step's logits rows are within 3e-2 of max|logit| of the oracle's next-token logits for that prefix."""
    from dataclasses import replace
    from llava_mi355x import _C
    from llava_mi355x.batching import DecodeBatch
    from llava_mi355x.model import LmxKVCache
    from oracle import llava_oracle as O
    from synthetic import build as harness, recipes as synth
    cfg = replace(synth.with_layers(synth.CONFIGS["llava15_7b"], 1, 1), init="unit", mm_vision_select_layer=-1, max_position_embeddings=1024)
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    dt, n_new = torch.bfloat16, 3
    n_img = 4                                                     # distinct images (the tower is not what is tested); every request has its own prompt
    pixs = [torch.from_numpy(synth.make_pixels(cfg, 1, seed=6 + i)) for i in range(n_img)]
    reqs = [(torch.from_numpy(synth.make_prompt(cfg, 20 + 5 * i, image_positions=(3 + i % 7,), seed=5 + i))[None], pixs[i % n_img]) for i in range(n_req)]
    model = harness.build_model(cfg, dtype=dt, weights=wnp)
    gen = model.generate_batch([i.cuda() for i, _ in reqs], [p.cuda().to(dt) for _, p in reqs], max_new_tokens=n_new, eos_token_id=-1, run_ahead=1)
    # the oracle on a few of the requests (a 7B-width layer + lm_head on 600+ rows costs seconds on the host): first, middle, last
    for r in sorted({0, n_req // 2, n_req - 1}):
        ids, pix = reqs[r]
        L = ids.shape[1]
        g = gen[r].cpu()
        assert g.shape[0] == L + n_new and g[:L].tolist() == ids[0].tolist()
        for t in range(n_new):
            with torch.no_grad():
                lg = O.llava_forward(w, cfg, g[None, : L + t], pix, last_only=True)[0][0, -1].float()
            tok = int(g[L + t])
            assert (lg.max() - lg[tok]).item() <= 3e-2 * lg.abs().max().item(), (r, t, tok, int(lg.argmax()))
    # logits of ONE batched step for all members against the oracle's next-token logits (members 0 and n_req - 1)
    caches = []
    for ids, pix in reqs:
        _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids.cuda(), None, None, None, None, pix.cuda().to(dt))
        c = LmxKVCache(model, 1)
        _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
        caches.append(c)
    bt = DecodeBatch(model, n_req)
    lb = torch.empty((n_req, cfg.vocab_size), dtype=dt, device=cuda)
    picks = bt.step([c.seqs[0] for c in caches], None, 1, True, lb)       # consumes each member's prefill pick, produces the next
    torch.cuda.synchronize()
    for r in (0, n_req - 1):
        ids, pix = reqs[r]
        first = int(gen[r][ids.shape[1]])                                   # the prefill's pick = the token this step consumed
        with torch.no_grad():
            lg = O.llava_forward(w, cfg, torch.cat([ids, torch.tensor([[first]])], 1), pix, last_only=True)[0][0, -1].float()
        err = (lb[r].float().cpu() - lg).abs().max().item() / lg.abs().max().item()
        assert err <= 3e-2, (r, err)
        assert (lg.max() - lg[picks[0][r]]).item() <= 3e-2 * lg.abs().max().item()
    bt.close()
    for c in caches:
        c.close()
