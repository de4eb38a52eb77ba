"""CPU, build container only: the entry glue of llava_mi355x/train_entry.py against the reference's own code — initialize_vision_tokenizer on the reference's
model class (llava/model/llava_arch.py:242-284), a fresh projector against build_vision_projector under the same seed (multimodal_projector/builder.py:33-51),
the adapter file round trip, the learning-rate schedule against transformers.get_scheduler."""
import copy
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available() or ref_shim.is_sourceless(), reason="reference source tree not present")


def _setup():
    from oracle import llava_oracle as O
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    lc, _ = harness.hf_configs(cfg)
    return cfg, wnp, lc


@pytest.mark.parametrize("tune", [False, True])
def test_initialize_vision_tokenizer_equal(tmp_path, tune):
    """<im_start> / <im_end>: two new rows = the mean of the old ones, in both matrices; with an adapter file the embedding rows come from it; the trainable flags."""
    from tok_util import build_tokenizer
    from llava_mi355x import train_entry as E
    cfg, wnp, lc = _setup()
    ref_model = ref_shim.build_reference_model(cfg, wnp)
    weights = {k: torch.from_numpy(np.array(v)) for k, v in wnp.items() if k in ("model.embed_tokens.weight", "lm_head.weight")}
    V0 = weights["model.embed_tokens.weight"].shape[0]
    adapter = None
    if tune:
        saved = {"model.embed_tokens.weight": torch.randn(2, cfg.hidden_size), "model.mm_projector.0.weight": torch.zeros(1)}
        adapter = str(tmp_path / "mm_projector.bin")
        torch.save(saved, adapter)
    args = types.SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=True, tune_mm_mlp_adapter=tune, pretrain_mm_mlp_adapter=adapter)
    tok_a, tok_b = build_tokenizer(vocab_size=V0), build_tokenizer(vocab_size=V0)
    assert len(tok_a) == V0
    got = E.initialize_vision_tokenizer(copy.deepcopy(lc), weights, args, tok_a)
    ref_model.initialize_vision_tokenizer(args, tokenizer=tok_b)
    assert got["num_new_tokens"] == 2 and len(tok_a) == len(tok_b) == V0 + 2
    re, rh = ref_model.get_input_embeddings().weight.data, ref_model.get_output_embeddings().weight.data
    assert weights["model.embed_tokens.weight"].shape == re.shape and weights["lm_head.weight"].shape == rh.shape
    assert torch.equal(weights["model.embed_tokens.weight"][:V0], re[:V0]) and torch.equal(weights["lm_head.weight"][:V0], rh[:V0])
    assert torch.allclose(weights["model.embed_tokens.weight"][V0:], re[V0:], atol=1e-7, rtol=0)
    assert torch.allclose(weights["lm_head.weight"][V0:], rh[V0:], atol=1e-7, rtol=0)
    if tune:
        assert torch.equal(weights["model.embed_tokens.weight"][V0:], saved["model.embed_tokens.weight"])
        assert got["embed_tokens_trainable"] is True and got["lm_head_trainable"] is False
        assert all(p.requires_grad for p in ref_model.get_input_embeddings().parameters()) and not any(p.requires_grad for p in ref_model.get_output_embeddings().parameters())
    else:
        assert got["embed_tokens_trainable"] is None and got["lm_head_trainable"] is None


def test_patch_token_only_resizes_and_freezes():
    from tok_util import build_tokenizer
    from llava_mi355x import train_entry as E
    cfg, wnp, lc = _setup()
    weights = {k: torch.from_numpy(np.array(wnp[k])) for k in ("model.embed_tokens.weight", "lm_head.weight")}
    V0 = weights["lm_head.weight"].shape[0]
    tok = build_tokenizer(vocab_size=V0)
    c = copy.deepcopy(lc)
    got = E.initialize_vision_tokenizer(c, weights, types.SimpleNamespace(mm_use_im_patch_token=True, mm_use_im_start_end=False, tune_mm_mlp_adapter=True,
                                                                         pretrain_mm_mlp_adapter=None), tok)
    assert c.vocab_size == V0 + 1 == weights["model.embed_tokens.weight"].shape[0] == weights["lm_head.weight"].shape[0]
    assert torch.equal(weights["lm_head.weight"][:V0], torch.from_numpy(np.array(wnp["lm_head.weight"])))
    assert got == {"num_new_tokens": 0, "embed_tokens_trainable": False, "lm_head_trainable": False}


@pytest.mark.parametrize("ptype", ["linear", "mlp2x_gelu", "mlp3x_gelu"])
def test_fresh_projector_equals_the_reference_builder_under_the_same_seed(tmp_path, ptype):
    from llava_mi355x import train_entry as E
    ref_shim.load_reference()
    from llava.model.multimodal_projector.builder import build_vision_projector
    cfg, wnp, lc = _setup()
    c = copy.deepcopy(lc)
    args = types.SimpleNamespace(vision_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                                 mm_projector_type=ptype, pretrain_mm_mlp_adapter=None)
    weights = {}
    torch.manual_seed(7)
    E.initialize_vision_modules(c, weights, args, vision_hidden_size=cfg.v_hidden_size)
    assert (c.mm_hidden_size, c.mm_projector_type, c.use_mm_proj, c.mm_vision_select_layer, c.mm_vision_tower) == (cfg.v_hidden_size, ptype, True, -2, args.vision_tower)
    torch.manual_seed(7)
    ref_proj = build_vision_projector(c)
    sd = ref_proj.state_dict()
    assert {k[len("model.mm_projector."):] for k in weights} == set(sd)
    for k, v in sd.items():
        assert torch.equal(weights["model.mm_projector." + k], v), k
    # the adapter file round trip: what save_checkpoint writes for the tune_mm_mlp_adapter stage is what pretrain_mm_mlp_adapter reads back
    state = {k.replace("model.mm_projector.", "mm_projector."): v + 1 for k, v in weights.items()}
    state["model.embed_tokens.weight"] = torch.zeros(4, 4)
    path = E.save_checkpoint(str(tmp_path), c, state, types.SimpleNamespace(tune_mm_mlp_adapter=True, mm_use_im_start_end=False))
    saved = torch.load(path)
    assert os.path.basename(path) == "mm_projector.bin" and set(saved) == set(weights)
    again = {}
    args.pretrain_mm_mlp_adapter = path
    E.initialize_vision_modules(copy.deepcopy(lc), again, args, vision_hidden_size=cfg.v_hidden_size)
    for k in weights:
        assert torch.equal(again[k], weights[k] + 1), k
    ref_proj.load_state_dict({k.split("mm_projector.")[1]: v for k, v in saved.items()})          # the reference's own loading expression (llava_arch.py:77-81)


@pytest.mark.parametrize("kind,ratio", [("cosine", 0.03), ("cosine", 0.0), ("linear", 0.1), ("constant_with_warmup", 0.25)])
def test_learning_rate_schedule_equals_transformers(kind, ratio):
    import math
    from transformers import get_scheduler
    from llava_mi355x.train_entry import lr_at
    total, base = 37, 2e-5
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=base)
    sch = get_scheduler(kind, opt, num_warmup_steps=math.ceil(total * ratio), num_training_steps=total)
    for step in range(total):
        assert abs(opt.param_groups[0]["lr"] - lr_at(step, total, base, ratio, kind)) <= 1e-12 * base + 1e-18, (kind, step)
        opt.step(); sch.step()


def test_smart_tokenizer_and_embedding_resize_equal():
    """`--version v0`: a [PAD] token for a tokenizer without one; both matrices grow by one row = the mean of the old rows — against the reference's function on its
    own model class."""
    from tok_util import build_tokenizer
    from llava_mi355x import train_entry as E
    ref = ref_shim.load_reference_train()
    cfg, wnp, lc = _setup()
    ref_model = ref_shim.build_reference_model(cfg, wnp)
    weights = {k: torch.from_numpy(np.array(wnp[k])) for k in ("model.embed_tokens.weight", "lm_head.weight")}
    V0 = weights["lm_head.weight"].shape[0]
    tok_a, tok_b = build_tokenizer(vocab_size=V0), build_tokenizer(vocab_size=V0)
    for t in (tok_a, tok_b):
        t.pad_token = None
    c = copy.deepcopy(lc)
    n = E.smart_tokenizer_and_embedding_resize(dict(pad_token="[PAD]"), tok_a, c, weights)
    ref.train.smart_tokenizer_and_embedding_resize(dict(pad_token="[PAD]"), tok_b, ref_model)
    assert n == 1 and len(tok_a) == len(tok_b) == V0 + 1 == c.vocab_size and tok_a.pad_token_id == tok_b.pad_token_id == V0
    for key, rw in (("model.embed_tokens.weight", ref_model.get_input_embeddings().weight.data), ("lm_head.weight", ref_model.get_output_embeddings().weight.data)):
        assert weights[key].shape == rw.shape and torch.equal(weights[key][:V0], rw[:V0])
        assert torch.allclose(weights[key][V0:], rw[V0:], atol=1e-7, rtol=0)
