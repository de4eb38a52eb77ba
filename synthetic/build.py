"""Stand up the PRODUCT model (LlavaLlamaForCausalLM on the MI355X engine) from the synthetic recipes: what bench.py, tools/ and
the tests use in place of load_pretrained_model when there is no checkpoint.  Contains no reference arithmetic (that is oracle/)."""
from __future__ import annotations

import os
import sys

import torch

from . import recipes as synth

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)


def hf_configs(cfg: synth.SynthConfig):
    """(LlavaConfig, CLIPVisionConfig) equivalent to a SynthConfig."""
    from transformers import CLIPVisionConfig
    from llava_mi355x.model import LlavaConfig
    lc = LlavaConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_position_embeddings,
                     pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    lc.rope_theta = cfg.rope_theta
    lc.mm_vision_tower = f"synthetic://{cfg.name}"
    lc.mm_projector_type = cfg.mm_projector_type
    lc.mm_hidden_size = cfg.v_hidden_size
    lc.mm_vision_select_layer = cfg.mm_vision_select_layer
    lc.mm_vision_select_feature = cfg.mm_vision_select_feature
    lc.tokenizer_padding_side = cfg.tokenizer_padding_side
    lc.tokenizer_model_max_length = cfg.tokenizer_model_max_length
    vc = CLIPVisionConfig(hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size, num_hidden_layers=cfg.v_num_hidden_layers,
                          num_attention_heads=cfg.v_num_attention_heads, image_size=cfg.v_image_size, patch_size=cfg.v_patch_size,
                          layer_norm_eps=cfg.v_layer_norm_eps, hidden_act="quick_gelu")
    return lc, vc


def build_model(cfg: synth.SynthConfig, dtype=torch.bfloat16, seed: int = 0, weights=None, device_rng: bool = False, round_weights_to=None, **kw):
    """Construct the product model and load synthetic weights tensor by tensor (bounded host memory at 7B scale).

    device_rng=True draws the weights with torch's device generator instead of numpy (benchmark-only: values differ from
    the numpy recipe, statistics are the same) so a 7B model materialises in seconds.  round_weights_to: every tensor is rounded to that dtype
    first — an fp32 model then holds exactly the values a 16-bit model of the same seed holds (the bench's live parity probe: fp32 verification engine vs the
    timed bf16 kernels on identical parameters)."""
    from llava_mi355x.model import LlavaLlamaForCausalLM
    lc, vc = hf_configs(cfg)
    model = LlavaLlamaForCausalLM(lc, vc, dtype=dtype, **kw)
    shapes = synth.tensor_shapes(cfg)
    gen = torch.Generator(device=model.device)
    gen.manual_seed(seed)
    for name, shp in shapes.items():
        if weights is not None:
            t = torch.from_numpy(weights[name])
        elif device_rng:
            t = _device_tensor(cfg, name, shp, gen, model.device)
        else:
            t = torch.from_numpy(synth.make_tensor(cfg, name, shp, seed))
        if round_weights_to is not None:
            t = t.to(round_weights_to).to(torch.float32)
        model.load_tensor(name, t)
    model.finalize_weights()
    model.get_vision_tower().is_loaded = True
    return model


def _device_tensor(cfg, name, shape, gen, device):
    is_norm_w = (name.endswith("norm.weight") or ("layer_norm" in name and name.endswith(".weight")) or name.endswith("pre_layrnorm.weight"))
    if is_norm_w:
        return torch.ones(shape, device=device)
    if name.endswith(".bias"):
        return torch.zeros(shape, device=device)
    return torch.randn(shape, device=device, generator=gen) * 0.02
