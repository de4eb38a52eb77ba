"""CPU: the reader of peft LoRA adapters behind load_pretrained_model's LoRA branch (llava/model/builder.py:50-81 -> PeftModel.from_pretrained):
key naming (`base_model.model.<module>.lora_A[.default].weight`), scaling rules, refused adapter kinds.  The merge itself runs on the device
(tests/test_loader_gpu.py::test_lora_checkpoint_on_base_llm)."""
import json
import os

import pytest
import torch


def _adapter(d, keys, cfg, fmt="bin"):
    os.makedirs(d, exist_ok=True)
    json.dump(cfg, open(os.path.join(d, "adapter_config.json"), "w"))
    if fmt == "bin":
        torch.save(keys, os.path.join(d, "adapter_model.bin"))
    else:
        from safetensors.torch import save_file
        save_file(keys, os.path.join(d, "adapter_model.safetensors"))


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
@pytest.mark.parametrize("infix", ["", ".default"])
def test_read_lora_adapter_names_and_scaling(tmp_path, fmt, infix):
    from llava_mi355x.builder import read_lora_adapter
    A, B = torch.randn(4, 32), torch.randn(48, 4)
    keys = {f"base_model.model.model.layers.3.mlp.up_proj.lora_A{infix}.weight": A, f"base_model.model.model.layers.3.mlp.up_proj.lora_B{infix}.weight": B}
    _adapter(str(tmp_path), keys, {"peft_type": "LORA", "r": 4, "lora_alpha": 10, "bias": "none"}, fmt)
    pairs, scaling = read_lora_adapter(str(tmp_path))
    assert list(pairs) == ["model.layers.3.mlp.up_proj.weight"] and scaling == 2.5
    assert torch.equal(pairs["model.layers.3.mlp.up_proj.weight"][0], A) and torch.equal(pairs["model.layers.3.mlp.up_proj.weight"][1], B)
    _adapter(str(tmp_path), keys, {"peft_type": "LORA", "r": 4, "lora_alpha": 10, "bias": "none", "use_rslora": True}, fmt)
    assert read_lora_adapter(str(tmp_path))[1] == 5.0


@pytest.mark.parametrize("bad", [{"peft_type": "IA3"}, {"bias": "all"}, {"fan_in_fan_out": True}, {"use_dora": True}, {"modules_to_save": ["lm_head"]}])
def test_read_lora_adapter_refuses_what_it_cannot_merge(tmp_path, bad):
    from llava_mi355x.builder import read_lora_adapter
    keys = {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(2, 8), "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": torch.zeros(8, 2)}
    _adapter(str(tmp_path), keys, {"peft_type": "LORA", "r": 2, "lora_alpha": 4, "bias": "none", **bad})
    with pytest.raises(NotImplementedError):
        read_lora_adapter(str(tmp_path))


def test_read_lora_adapter_shape_and_pair_checks(tmp_path):
    from llava_mi355x.builder import read_lora_adapter
    cfg = {"peft_type": "LORA", "r": 2, "lora_alpha": 4, "bias": "none"}
    _adapter(str(tmp_path / "a"), {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(2, 8)}, cfg)
    with pytest.raises(ValueError):
        read_lora_adapter(str(tmp_path / "a"))
    _adapter(str(tmp_path / "b"), {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": torch.zeros(2, 8),
                                   "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": torch.zeros(8, 3)}, cfg)
    with pytest.raises(ValueError):
        read_lora_adapter(str(tmp_path / "b"))
    _adapter(str(tmp_path / "c"), {"something.else": torch.zeros(2)}, cfg)
    with pytest.raises(ValueError):
        read_lora_adapter(str(tmp_path / "c"))
    os.makedirs(str(tmp_path / "d"))
    json.dump(cfg, open(str(tmp_path / "d" / "adapter_config.json"), "w"))
    with pytest.raises(FileNotFoundError):
        read_lora_adapter(str(tmp_path / "d"))


def test_from_pretrained_applies_overrides_then_lora(tmp_path, monkeypatch):
    """The loading loop of builder.from_pretrained on the CPU (engine model and device merge replaced by recorders): a tensor of `overrides`
    (non_lora_trainables.bin) replaces the checkpoint's, LoRA deltas are added to what is loaded (override included — the reference loads the extra
    state dict first and merges afterwards, builder.py:74-80), overrides the checkpoint does not hold are loaded too, vision / skipped keys stay out,
    and an adapter that targets a weight nobody holds is an error."""
    import types
    from safetensors.torch import save_file
    from llava_mi355x import builder
    from llava_mi355x.model import LlavaLlamaForCausalLM as Real

    loaded = {}

    class Fake:
        canonical_name = staticmethod(Real.canonical_name)
        device = "cpu"

        def __init__(self, *a, **k):
            pass

        def load_tensor(self, name, t):
            loaded[name] = t.clone()

    monkeypatch.setattr(builder, "LlavaLlamaForCausalLM", Fake)
    monkeypatch.setattr(builder, "merge_lora", lambda W, A, B, s, dev: W.float() + (B.float() @ A.float()) * s)
    import transformers
    monkeypatch.setattr(transformers.CLIPVisionConfig, "from_pretrained", classmethod(lambda cls, p, **k: object()))
    ck = {"model.layers.0.self_attn.q_proj.weight": torch.ones(4, 4), "model.layers.0.mlp.up_proj.weight": torch.full((6, 4), 2.0),
          "model.norm.weight": torch.ones(4), "model.vision_tower.vision_tower.vision_model.x": torch.zeros(1), "model.layers.0.self_attn.rotary_emb.inv_freq": torch.zeros(2)}
    save_file(ck, str(tmp_path / "model.safetensors"))
    cfg = types.SimpleNamespace(mm_vision_tower="unused", mm_vision_select_layer=-2)
    A, B = torch.randn(2, 4), torch.randn(6, 2)
    lora = ({"model.layers.0.mlp.up_proj.weight": (A, B)}, 0.5)
    over = {"model.layers.0.mlp.up_proj.weight": torch.full((6, 4), 3.0), "model.mm_projector.0.weight": torch.full((4, 4), 7.0)}
    builder.from_pretrained(str(tmp_path), config=cfg, torch_dtype=torch.float32, device="cpu", overrides=over, lora=lora)
    assert set(loaded) == {"model.layers.0.self_attn.q_proj.weight", "model.layers.0.mlp.up_proj.weight", "model.norm.weight", "mm_projector.0.weight"}
    assert torch.equal(loaded["model.layers.0.mlp.up_proj.weight"], torch.full((6, 4), 3.0) + (B @ A) * 0.5)
    assert torch.equal(loaded["mm_projector.0.weight"], torch.full((4, 4), 7.0)) and torch.equal(loaded["model.layers.0.self_attn.q_proj.weight"], torch.ones(4, 4))
    loaded.clear()
    with pytest.raises(KeyError):
        builder.from_pretrained(str(tmp_path), config=cfg, torch_dtype=torch.float32, device="cpu",
                                lora=({"model.layers.9.mlp.up_proj.weight": (A, B)}, 0.5))
