"""A whole visual-instruction-tuning step (llava_mi355x/train.py; SURVEY §8 f-3, BASELINE config 5) against torch autograd +
torch.optim.AdamW on the oracle.

The reference step: HF Trainer over LlavaLlamaForCausalLM.forward(labels=...) (llava/train/train.py:805-1000, llava_llama.py:56-99) with the CLIP
tower frozen (clip_encoder.py:25), LLM + mm_projector trainable, AdamW, max_grad_norm clipping, ZeRO-2 (scripts/zero2.json).  Here: a batch of
two samples of different lengths (right-padded, one image each), tiny geometries with the mlp2x_gelu and the linear projector (GQA), fp32 for the
tight comparison:
    loss                      <= 1e-5 relative
    every gradient tensor     <= 1e-3 of max|ref| (fp32 atomics reorder the dK / dV / embedding sums; GEMM accumulation order differs)
    clipped global grad norm  <= 1e-4 relative
    parameters after AdamW    elements with a non-negligible gradient move by the reference's update (1e-2 relative); the others (whose
                              first-step Adam direction g / |g| is decided by rounding noise) by at most one lr
and bf16 for the rounding-chain run: loss within 2e-2, per-tensor gradient cosine >= 0.98.  checkpoint=True (recompute) must reproduce the
stored-activation gradients, and the ZeRO-2 partition with two ranks on this GPU must reproduce the single-rank parameters."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def make_batch(cfg, seed=0):
    from synthetic import recipes as synth
    rng = np.random.RandomState(seed)
    L = 22
    a = synth.make_prompt(cfg, L, image_positions=(4,), seed=seed + 2)
    b = synth.make_prompt(cfg, L, image_positions=(7,), seed=seed + 3)
    ids = torch.from_numpy(np.stack([a, b]))
    mask = torch.ones_like(ids); mask[1, 15:] = 0                      # second sample is shorter (right padding)
    labels = ids.clone(); labels[0, :9] = -100; labels[1, :11] = -100    # prompt part masked, as preprocess_* does (train.py:254-638)
    labels[ids == -200] = -100
    pix = torch.from_numpy(synth.make_pixels(cfg, 2, seed=seed + 1))
    return ids, mask, labels, pix


def trainable(name):
    return not name.startswith("vision.") and "vision_tower" not in name


def reference_step(cfg, wnp, batch, lr, wd, max_norm, steps=1):
    """torch autograd over the oracle + clip_grad_norm_ + torch.optim.AdamW (what HF Trainer does)."""
    from oracle import llava_oracle as O
    ids, mask, labels, pix = batch
    w = O.to_torch_weights(wnp)
    params = {k: v.clone().requires_grad_(True) for k, v in w.items() if trainable(k)}
    full = dict(w); full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    out = {}
    for _ in range(steps):
        opt.zero_grad()
        logits, _, _, new_labels = O.llava_forward(full, cfg, ids, pix, attention_mask=mask, labels=labels)
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, cfg.vocab_size), new_labels[:, 1:].reshape(-1), ignore_index=-100)
        loss.backward()
        out.setdefault("loss", []).append(loss.item())
        if "grads" not in out:
            out["grads"] = {k: v.grad.detach().clone() for k, v in params.items()}
        norm = torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm)
        out.setdefault("norm", []).append(float(norm))
        opt.step()
    out["params"] = {k: v.detach().clone() for k, v in params.items()}
    with torch.no_grad():
        out["tower"] = O.vision_tower(w, cfg, pix)
    return out


def build_step(cfg, wnp, dtype, device, **kw):
    from llava_mi355x.train import TrainStep
    from synthetic import build as harness
    lc, _ = harness.hf_configs(cfg)
    weights = {k: torch.from_numpy(v) for k, v in wnp.items() if trainable(k)}
    return TrainStep(lc, weights, dtype=dtype, device=device, **kw)


def _rel(got, ref):
    return ((got.detach().float().cpu() - ref.float()).abs().max() / ref.float().abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
def test_one_step_matches_autograd_and_adamw_fp32(cuda, name):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg)
    lr, wd, max_norm = 1e-3, 0.0, 1.0
    ref = reference_step(cfg, wnp, batch, lr, wd, max_norm)
    ts = build_step(cfg, wnp, torch.float32, cuda, lr=lr, weight_decay=wd, max_grad_norm=max_norm)
    ids, mask, labels, _ = batch
    loss, count = ts.forward_backward(ids, labels, mask, image_features=ref["tower"])
    assert int(count.item()) == int((labels[:, 1:] != -100).logical_and(mask[:, 1:].bool()).sum())
    assert abs(loss.item() - ref["loss"][0]) <= 1e-5 * abs(ref["loss"][0])
    worst = {}
    for k, g in ref["grads"].items():
        kk = k.replace("model.mm_projector.", "mm_projector.")
        worst[k] = _rel(ts.g[kk], g)
    bad = {k: v for k, v in worst.items() if v > 1e-3}
    assert not bad, bad
    before = ts.state_dict()
    ts.optimizer_step()
    assert abs(ts.grad_norm() - ref["norm"][0]) <= 1e-4 * ref["norm"][0]
    w0 = {k: torch.from_numpy(v) for k, v in wnp.items()}
    for k, p_ref in ref["params"].items():
        kk = k.replace("model.mm_projector.", "mm_projector.")
        d_ref = p_ref - w0[k]
        d_got = ts.p[kk].detach().cpu() - before[kk].cpu()
        g = ref["grads"][k]
        solid = g.abs() > 1e-3 * g.abs().max()
        if solid.any():
            assert (d_got[solid] - d_ref[solid]).abs().max().item() <= 1e-2 * lr, k
        assert (d_got - d_ref).abs().max().item() <= 2.001 * lr, k
        assert float(d_got[g == 0].abs().max() if (g == 0).any() else 0.0) == 0.0, k            # untouched rows (unused token ids) do not move


def test_three_steps_track_the_reference_fp32(cuda):
    """Loss trajectory over three optimisation steps on the same batch (moments, bias correction and the master copy all in play)."""
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg, seed=5)
    ref = reference_step(cfg, wnp, batch, 5e-4, 0.01, 1.0, steps=3)
    ts = build_step(cfg, wnp, torch.float32, cuda, lr=5e-4, weight_decay=0.01, max_grad_norm=1.0)
    ids, mask, labels, _ = batch
    losses = []
    for _ in range(3):
        loss, _ = ts.step(ids, labels, mask, image_features=ref["tower"])
        losses.append(loss.item())
    assert ref["loss"][2] < ref["loss"][0]                                   # it does learn
    for a, b in zip(losses, ref["loss"]):
        assert abs(a - b) <= 2e-3 * abs(b), (losses, ref["loss"])


def test_checkpointing_reproduces_gradients(cuda):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg)
    ref = reference_step(cfg, wnp, batch, 1e-3, 0.0, 1.0)
    ids, mask, labels, _ = batch
    a = build_step(cfg, wnp, torch.float32, cuda)
    b = build_step(cfg, wnp, torch.float32, cuda, checkpoint=True)
    la, _ = a.forward_backward(ids, labels, mask, image_features=ref["tower"])
    lb, _ = b.forward_backward(ids, labels, mask, image_features=ref["tower"])
    assert la.item() == lb.item()
    for k in a.g:
        assert _rel(b.g[k], a.g[k].cpu()) <= 1e-5, k


def test_kept_layers_reproduce_recomputed_gradients(cuda):
    """Round 6: with checkpoint=True the last `keep_layers` layers (or as many as `keep_budget_bytes` holds) keep their activations and are not recomputed —
    a memory / time trade of the 288-GB part, not an arithmetic change: loss and every gradient are the SAME BITS as with full recomputation, whatever the split."""
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg)
    ref = reference_step(cfg, wnp, batch, 1e-3, 0.0, 1.0)
    ids, mask, labels, _ = batch
    for dt in (torch.float32, torch.bfloat16):
        full = build_step(cfg, wnp, dt, cuda, checkpoint=True)
        l0, _ = full.forward_backward(ids, labels, mask, image_features=ref["tower"])
        assert full.last_kept_layers == 0
        for kw, want in ((dict(keep_layers=1), 1), (dict(keep_layers=99), full.L), (dict(keep_budget_bytes=0), 0), (dict(keep_budget_bytes=1 << 40), full.L)):
            part = build_step(cfg, wnp, dt, cuda, checkpoint=True, **kw)
            l1, _ = part.forward_backward(ids, labels, mask, image_features=ref["tower"])
            assert part.last_kept_layers == want, (kw, part.last_kept_layers)
            assert l1.item() == l0.item()
            for k in full.g:
                assert torch.equal(part.g[k], full.g[k]), (kw, k)
        one = build_step(cfg, wnp, dt, cuda, checkpoint=True, keep_budget_bytes=full.layer_activation_bytes(64) * 1)
        one.forward_backward(ids, labels, mask, image_features=ref["tower"])
        assert one.last_kept_layers in (0, 1)                      # the packed rows of this batch fill at least one quantum of 64


def test_one_step_bf16(cuda):
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    batch = make_batch(cfg)
    ref = reference_step(cfg, wnp, batch, 1e-3, 0.0, 1.0)
    ts = build_step(cfg, wnp, torch.bfloat16, cuda, lr=1e-3)
    ids, mask, labels, _ = batch
    loss, _ = ts.forward_backward(ids, labels, mask, image_features=ref["tower"])
    assert abs(loss.item() - ref["loss"][0]) <= 2e-2 * abs(ref["loss"][0])
    for k, g in ref["grads"].items():
        kk = k.replace("model.mm_projector.", "mm_projector.")
        got = ts.g[kk].float().cpu().flatten(); want = g.flatten()
        cos = float((got @ want) / (got.norm() * want.norm()).clamp_min(1e-20))
        assert cos >= 0.98, (k, cos)
    ts.optimizer_step()
    assert torch.isfinite(ts.flat_p.float()).all()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_zero2_two_ranks_match_single_rank(cuda, tmp_path):
    """Two processes = two data-parallel ranks on this GPU (gloo group, buckets staged through the host).  Same micro-batch on both ranks:
    the averaged gradients equal the single-rank ones, so after reduce-scatter -> clip -> AdamW on the owned slices -> all-gather every rank
    must hold the single-rank parameters.  Different micro-batches: ranks agree with each other."""
    port = _free_port()
    procs = []
    for r in range(2):
        out = str(tmp_path / f"z{r}.pt")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "zero2_worker.py"), str(r), "2", str(port), out], stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT), out))
    logs = []
    for p, _ in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
        logs.append(o.decode(errors="replace")[-2500:])
    res = []
    for (_, out), lg in zip(procs, logs):
        assert os.path.exists(out), lg
        res.append(torch.load(out))
    for r in res:
        assert r["ok"], r.get("trace")
    assert res[0]["n_buckets"] > 3 and res[0]["shard_elems"] * 2 == res[0]["total"]
    assert torch.equal(res[0]["same_batch_params"], res[1]["same_batch_params"])
    assert torch.equal(res[0]["diff_batch_params"], res[1]["diff_batch_params"])
    single = res[0]["single_params"]
    # run-to-run the fp32 atomics (dK / dV / embedding sums) move gradients in the last bits; at step 1 Adam's direction g / |g| turns that into
    # an lr-sized move for the few elements whose gradient is rounding noise, so: nearly all elements identical, none further than 2 lr
    diff = (res[0]["same_batch_params"] - single).abs()
    assert diff.max().item() <= 2.001e-3 and (diff > 1e-6).float().mean().item() < 1e-2
    other = (res[0]["diff_batch_params"] - single).abs()
    assert (other > 1e-5).float().mean().item() > 0.05                                  # the other rank's batch really took part
    assert abs(res[0]["same_norm"] - res[0]["single_norm"]) <= 1e-5 * res[0]["single_norm"]


@pytest.mark.parametrize("case", ["left_padding", "two_images_one_row", "truncation"])
def test_step_edge_cases_of_the_splice(cuda, case):
    """The training batch goes through the same splice plan as inference (llava_arch.py:99-240): left padding (tokenizer_padding_side), a sample with
    two <image> markers, truncation to tokenizer_model_max_length AFTER the image expansion — loss and every gradient must still match autograd
    over the oracle (whose splice is pinned to the reference's by the golden cases of the same names)."""
    from dataclasses import replace
    from oracle import llava_oracle as O
    from synthetic import build as harness, recipes as synth
    from llava_mi355x.train import TrainStep
    base = synth.CONFIGS["tiny"]
    cfg = base
    L = 22
    a = synth.make_prompt(base, L, image_positions=(4,), seed=21)
    b = synth.make_prompt(base, L, image_positions=(7,), seed=22)
    n_img = 2
    if case == "left_padding":
        cfg = replace(base, tokenizer_padding_side="left")
    elif case == "two_images_one_row":
        b = synth.make_prompt(base, L, image_positions=(3, 12), seed=23); n_img = 3
    else:
        cfg = replace(base, tokenizer_model_max_length=30)
    ids = torch.from_numpy(np.stack([a, b]))
    mask = torch.ones_like(ids); mask[0, 17:] = 0
    labels = ids.clone(); labels[:, :6] = -100; labels[ids == -200] = -100
    pix = torch.from_numpy(synth.make_pixels(base, n_img, seed=24))
    wnp = synth.make_weights(base, 0)
    w = O.to_torch_weights(wnp)
    params = {k: v.clone().requires_grad_(True) for k, v in w.items() if trainable(k)}
    full = dict(w); full.update(params)
    logits, _, _, new_labels = O.llava_forward(full, cfg, ids, pix, attention_mask=mask, labels=labels)
    loss = F.cross_entropy(logits[:, :-1].reshape(-1, cfg.vocab_size), new_labels[:, 1:].reshape(-1), ignore_index=-100)
    loss.backward()
    with torch.no_grad():
        tower = O.vision_tower(w, cfg, pix)
    lc, _ = harness.hf_configs(cfg)
    ts = TrainStep(lc, {k: torch.from_numpy(v) for k, v in wnp.items() if trainable(k)}, dtype=torch.float32, device=cuda)
    got, count = ts.forward_backward(ids, labels, mask, image_features=tower)
    assert int(count.item()) == int((new_labels[:, 1:] != -100).sum())
    assert abs(got.item() - loss.item()) <= 1e-5 * abs(loss.item())
    for k, p in params.items():
        assert _rel(ts.g[k], p.grad) <= 1e-3, (case, k)
