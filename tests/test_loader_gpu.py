"""load_pretrained_model (reference signature, llava/model/builder.py:26-151) on a self-generated HF-format checkpoint:
sharded safetensors with the key names the reference saves (`model.layers.*`, `model.mm_projector.*`, `lm_head.weight`),
`config.json` with the llava fields, a CLIP tower directory in the transformers-4.31 (`vision_model.*`) or 5.x (bare) key
layout, a sentencepiece LLaMA tokenizer.  The loaded model must reproduce the logits of a model built directly from the
same tensors, and a projector-only checkpoint on a base LLM (`mm_projector.bin`, builder.py:82-99) must load too."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from ckpt_util import write_clip as _write_clip, write_llava as _write_llava, write_tokenizer as _write_tokenizer  # noqa: E402


@pytest.mark.parametrize("layout", ["4.31", "5.x"])
def test_load_pretrained_model_roundtrip(cuda, tmp_path, layout):
    from llava_mi355x.builder import load_pretrained_model
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    clip_dir = str(tmp_path / "clip-tiny")
    _write_clip(clip_dir, cfg, wnp, layout)
    ckpt = str(tmp_path / "llava-tiny-7b")
    _write_llava(ckpt, cfg, wnp, clip_dir)
    tokenizer, model, image_processor, context_len = load_pretrained_model(ckpt, None, "llava-tiny-7b", torch_dtype=torch.float32)
    assert context_len == 2048 and image_processor is not None and tokenizer.bos_token_id == 1
    assert model.get_vision_tower().is_loaded and model.get_vision_tower().num_patches == cfg.num_patches
    direct = harness.build_model(cfg, dtype=torch.float32, weights=wnp)
    # builder.py:138: resize_token_embeddings(len(tokenizer)) — the vocabulary is now the tokenizer's (300 ids), like the reference's
    n_vocab = len(tokenizer)
    assert model.config.vocab_size == n_vocab < cfg.vocab_size
    ids_np = synth.make_prompt(cfg, 12, image_positions=(5,))
    ids_np[ids_np >= 0] %= n_vocab
    ids = torch.from_numpy(ids_np)[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1)).to(cuda)
    a = model.forward(input_ids=ids, images=pix, use_cache=False).logits
    b = direct.forward(input_ids=ids, images=pix, use_cache=False).logits
    assert a.shape[-1] == n_vocab and torch.equal(a, b[..., :n_vocab])
    with pytest.raises(IndexError):                      # an id beyond the resized vocabulary is an embedding index error, as upstream
        model.forward(input_ids=torch.full_like(ids, n_vocab + 1), use_cache=False)
    gen = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=8, eos_token_id=-1)
    assert int(gen[0, ids.shape[1]:].max()) < n_vocab    # padded / removed ids are never picked
    # the host helpers the worker uses work against the returned tokenizer (llava/serve/model_worker.py:133-171)
    from llava_mi355x.mm_utils import tokenizer_image_token
    tok = tokenizer_image_token("w1 w2 <image>\nw3", tokenizer, return_tensors="pt")
    assert tok[0].item() == 1 and (tok == -200).sum().item() == 1


def test_projector_only_checkpoint_on_base_llm(cuda, tmp_path):
    from llava_mi355x.builder import load_pretrained_model
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    clip_dir = str(tmp_path / "clip-tiny")
    _write_clip(clip_dir, cfg, wnp, "4.31")
    base = str(tmp_path / "vicuna-tiny")
    _write_llava(base, cfg, wnp, clip_dir, with_projector=False)
    proj = str(tmp_path / "llava-tiny-pretrain")
    os.makedirs(proj)
    for f in ("config.json",):
        open(os.path.join(proj, f), "w").write(open(os.path.join(base, f)).read())
    torch.save({"model." + k: torch.from_numpy(v) for k, v in wnp.items() if k.startswith("mm_projector.")}, os.path.join(proj, "mm_projector.bin"))
    tokenizer, model, image_processor, _ = load_pretrained_model(proj, base, "llava-tiny-pretrain", torch_dtype=torch.float32)
    direct = harness.build_model(cfg, dtype=torch.float32, weights=wnp)
    pix = torch.from_numpy(synth.make_pixels(cfg, 2)).to(cuda)
    assert torch.equal(model.encode_images(pix), direct.encode_images(pix))


def test_unsupported_requests_raise(cuda, tmp_path):
    from llava_mi355x.builder import load_pretrained_model
    for kw, name in ((dict(load_8bit=True), "llava-x"), (dict(load_4bit=True), "llava-x"), ({}, "llava-mpt-7b"), ({}, "llava-lora-x")):
        with pytest.raises(NotImplementedError):
            load_pretrained_model(str(tmp_path), None, name, **kw)
