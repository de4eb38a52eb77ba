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


def _write_lora_dir(d, base, cfg, wnp, r, alpha, fmt, seed=3):
    """What the reference's LoRA training leaves behind (llava/train/train.py:861-878 + peft save_pretrained): config.json, adapter_config.json,
    adapter_model.{safetensors,bin} with `base_model.model.<module>.lora_{A,B}.weight` keys, non_lora_trainables.bin (the projector)."""
    from safetensors.torch import save_file
    os.makedirs(d)
    open(os.path.join(d, "config.json"), "w").write(open(os.path.join(base, "config.json")).read())
    targets = ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "lora_dropout": 0.05, "bias": "none", "fan_in_fan_out": False, "target_modules": targets,
               "task_type": "CAUSAL_LM", "base_model_name_or_path": base}, open(os.path.join(d, "adapter_config.json"), "w"))
    g = torch.Generator().manual_seed(seed)
    ad, pairs = {}, {}
    for k, v in wnp.items():
        if not (k.startswith("model.layers.") and k.endswith(".weight") and k.split(".")[-2] in targets):
            continue
        out_f, in_f = v.shape
        A = torch.randn(r, in_f, generator=g) * 0.05
        B = torch.randn(out_f, r, generator=g) * 0.05
        mod = k[: -len(".weight")]
        ad[f"base_model.model.{mod}.lora_A.weight"] = A
        ad[f"base_model.model.{mod}.lora_B.weight"] = B
        pairs[k] = (A, B)
    if fmt == "safetensors":
        save_file(ad, os.path.join(d, "adapter_model.safetensors"))
    else:
        torch.save(ad, os.path.join(d, "adapter_model.bin"))
    torch.save({"base_model.model.model." + k: torch.from_numpy(v) for k, v in wnp.items() if k.startswith("mm_projector.")}, os.path.join(d, "non_lora_trainables.bin"))
    return pairs


@pytest.mark.parametrize("r,alpha,fmt", [(8, 16, "safetensors"), (12, 6, "bin")])
def test_lora_checkpoint_on_base_llm(cuda, tmp_path, r, alpha, fmt):
    """Un-merged LoRA checkpoint + base LLM (builder.py:50-81: base weights, non_lora_trainables.bin, PeftModel.from_pretrained, merge_and_unload): the loader
    merges W + (B @ A) * lora_alpha / r while it loads (fp32 on the device through the engine's GEMM).  The loaded model must reproduce a model built
    directly from weights merged in float64; rank not a multiple of 16 and a non-power-of-two scaling included."""
    from llava_mi355x.builder import load_pretrained_model
    from synthetic import build as harness
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    clip_dir = str(tmp_path / "clip-tiny")
    _write_clip(clip_dir, cfg, wnp, "4.31")
    base = str(tmp_path / "vicuna-tiny")
    _write_llava(base, cfg, wnp, clip_dir, with_projector=False)
    lora = str(tmp_path / "llava-tiny-lora")
    pairs = _write_lora_dir(lora, base, cfg, wnp, r, alpha, fmt)
    assert len(pairs) == 7 * cfg.num_hidden_layers
    tokenizer, model, _, _ = load_pretrained_model(lora, base, "llava-tiny-lora", torch_dtype=torch.float32)
    merged = dict(wnp)
    for k, (A, B) in pairs.items():
        merged[k] = (wnp[k].astype(np.float64) + (B.double() @ A.double()).numpy() * (alpha / r)).astype(np.float32)
    direct = harness.build_model(cfg, dtype=torch.float32, weights=merged)
    plain = harness.build_model(cfg, dtype=torch.float32, weights=wnp)
    n_vocab = len(tokenizer)
    ids_np = synth.make_prompt(cfg, 12, image_positions=(5,))
    ids_np[ids_np >= 0] %= n_vocab
    ids = torch.from_numpy(ids_np)[None].to(cuda)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1)).to(cuda)
    a = model.forward(input_ids=ids, images=pix, use_cache=False).logits
    b = direct.forward(input_ids=ids, images=pix, use_cache=False).logits[..., :n_vocab]
    c = plain.forward(input_ids=ids, images=pix, use_cache=False).logits[..., :n_vocab]
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-4 * scale                       # fp32 merge vs float64 merge rounded to fp32
    assert float((b - c).abs().max()) >= 1e-2 * scale                       # the adapter really changes the model
    assert torch.equal(model.encode_images(pix), direct.encode_images(pix))  # the projector came from non_lora_trainables.bin
    # 16-bit model: ONE rounding of the merged weight
    _, m16, _, _ = load_pretrained_model(lora, base, "llava-tiny-lora", torch_dtype=torch.bfloat16)
    d16 = harness.build_model(cfg, dtype=torch.bfloat16, weights=merged)
    a16 = m16.forward(input_ids=ids, images=pix.bfloat16(), use_cache=False).logits
    b16 = d16.forward(input_ids=ids, images=pix.bfloat16(), use_cache=False).logits[..., :n_vocab]
    assert float((a16 - b16).abs().max()) <= 2e-2 * float(b16.abs().max())   # a merged element may round the other way where fp32 and float64 sums straddle a bf16 tie


def test_lora_adapter_errors(cuda, tmp_path):
    from llava_mi355x.builder import load_pretrained_model, read_lora_adapter
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    clip_dir = str(tmp_path / "clip-tiny")
    _write_clip(clip_dir, cfg, wnp, "4.31")
    base = str(tmp_path / "vicuna-tiny")
    _write_llava(base, cfg, wnp, clip_dir, with_projector=False)
    lora = str(tmp_path / "llava-tiny-lora")
    _write_lora_dir(lora, base, cfg, wnp, 8, 16, "bin")
    ad = torch.load(os.path.join(lora, "adapter_model.bin"))
    ad["base_model.model.model.layers.99.self_attn.q_proj.lora_A.weight"] = torch.zeros(8, cfg.hidden_size)
    ad["base_model.model.model.layers.99.self_attn.q_proj.lora_B.weight"] = torch.zeros(cfg.hidden_size, 8)
    torch.save(ad, os.path.join(lora, "adapter_model.bin"))
    with pytest.raises(KeyError):                         # a target the base checkpoint does not hold is an error, not silently dropped
        load_pretrained_model(lora, base, "llava-tiny-lora", torch_dtype=torch.float32)
    cfgp = os.path.join(lora, "adapter_config.json")
    ac = json.load(open(cfgp)); ac["bias"] = "all"; json.dump(ac, open(cfgp, "w"))
    with pytest.raises(NotImplementedError):
        read_lora_adapter(lora)


def test_unsupported_requests_raise(cuda, tmp_path):
    from llava_mi355x.builder import load_pretrained_model
    for kw, name in ((dict(load_8bit=True), "llava-x"), (dict(load_4bit=True), "llava-x"), ({}, "llava-mpt-7b")):
        with pytest.raises(NotImplementedError):
            load_pretrained_model(str(tmp_path), None, name, **kw)
