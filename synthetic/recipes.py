"""Deterministic synthetic configs / weights / inputs for the LLaVA forward path.

Workload generator, neither oracle nor product: pure numpy recipes for configs, weights, pixels and prompts.  bench.py, tests/,
tools/ and the oracle all draw their inputs from here the product (llava-plus-codebase_amd/)
never imports it.

No real checkpoints, tokenizers or datasets exist in the build environment, so every parity check runs
on seeded synthetic tensors.  numpy's legacy RandomState is bit-stable across numpy/torch versions and platforms, so
goldens only need to store *outputs*: weights and inputs are regenerated from (config name, seed).

Canonical tensor names = HF checkpoint names with the llava prefixes stripped (the names lmx_load_weight accepts):
  model.embed_tokens.weight, model.layers.N.self_attn.{q,k,v,o}_proj.weight, model.layers.N.mlp.{gate,up,down}_proj.weight,
  model.layers.N.{input,post_attention}_layernorm.weight, model.norm.weight, lm_head.weight,
  mm_projector.{0,2}.{weight,bias},
  vision.embeddings.{class_embedding,patch_embedding.weight,position_embedding.weight}, vision.pre_layrnorm.{weight,bias},
  vision.encoder.layers.N.{self_attn.{q,k,v,out}_proj,mlp.{fc1,fc2},layer_norm1,layer_norm2}.{weight,bias}
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict

import numpy as np

IMAGE_TOKEN_INDEX = -200   # llava/constants.py:8
IGNORE_INDEX = -100        # llava/constants.py:7
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass(frozen=True)
class SynthConfig:
    name: str
    # LLaMA decoder (LlamaConfig fields)
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    # CLIP vision tower (CLIPVisionConfig fields)
    v_hidden_size: int = 1024
    v_intermediate_size: int = 4096
    v_num_hidden_layers: int = 24
    v_num_attention_heads: int = 16
    v_image_size: int = 336
    v_patch_size: int = 14
    v_layer_norm_eps: float = 1e-5
    # LLaVA glue (llava/model/llava_arch.py:48-68)
    mm_projector_type: str = "mlp2x_gelu"
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    tokenizer_padding_side: str = "right"
    tokenizer_model_max_length: int | None = None
    init: str = "unit"          # "unit": std 1/sqrt(fan_in) (O(1) activations, peaky attention) | "hf": std 0.02

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_patches(self) -> int:
        return (self.v_image_size // self.v_patch_size) ** 2

    @property
    def tokens_per_image(self) -> int:
        return self.num_patches if self.mm_vision_select_feature == "patch" else self.num_patches + 1

    @property
    def projector_depth(self) -> int:
        if self.mm_projector_type == "linear":
            return 1
        if self.mm_projector_type == "identity":
            return 0
        import re
        m = re.match(r"^mlp(\d+)x_gelu$", self.mm_projector_type)
        if not m:
            raise ValueError(f"Unknown projector type: {self.mm_projector_type}")
        return int(m.group(1))


CONFIGS: Dict[str, SynthConfig] = {
    # small enough for the CPU oracle to finish in well under a second; exercises every kernel family
    "tiny": SynthConfig("tiny", 256, 512, 2, 2, 2, 512, max_position_embeddings=256,
                        v_hidden_size=128, v_intermediate_size=256, v_num_hidden_layers=3, v_num_attention_heads=2,
                        v_image_size=56, v_patch_size=14),
    # grouped-query attention, head_dim 64, linear projector, 224px-style geometry (LLaVA-Plus v0 uses the linear projector)
    "tiny_gqa": SynthConfig("tiny_gqa", 256, 384, 2, 4, 2, 320, max_position_embeddings=256,
                            v_hidden_size=128, v_intermediate_size=256, v_num_hidden_layers=2, v_num_attention_heads=2,
                            v_image_size=42, v_patch_size=14, mm_projector_type="linear", mm_vision_select_layer=-1,
                            mm_vision_select_feature="cls_patch"),
    # the two remaining projector types of build_vision_projector (multimodal_projector/builder.py:33-51): identity (mm_hidden_size ==
    # hidden_size) and a deeper mlpNx_gelu
    "tiny_identity": SynthConfig("tiny_identity", 256, 512, 2, 2, 2, 512, max_position_embeddings=256,
                                 v_hidden_size=256, v_intermediate_size=512, v_num_hidden_layers=2, v_num_attention_heads=2,
                                 v_image_size=42, v_patch_size=14, mm_projector_type="identity"),
    "tiny_mlp3x": SynthConfig("tiny_mlp3x", 256, 384, 2, 4, 4, 320, max_position_embeddings=256,
                              v_hidden_size=128, v_intermediate_size=256, v_num_hidden_layers=2, v_num_attention_heads=2,
                              v_image_size=42, v_patch_size=14, mm_projector_type="mlp3x_gelu", mm_vision_select_layer=-1),
    # real LLaVA-1.5 geometry (BASELINE.json configs[0..2]); weights use HF init (std 0.02) as BASELINE.md prescribes
    "llava15_7b": SynthConfig("llava15_7b", 4096, 11008, 32, 32, 32, 32000, init="hf"),
    "llava15_13b": SynthConfig("llava15_13b", 5120, 13824, 40, 40, 40, 32000, init="hf"),
    # LLaVA-Plus v0 (BASELINE config 4 family): openai/clip-vit-large-patch14 at 224 px -> 256 patches, default LINEAR projector
    # (scripts/llava_plus/training_llava_plus_v0_7b.sh:18, no --mm_projector_type => llava/train/train.py:68)
    # BASELINE config 4's model: LLaVA-Plus on Vicuna-13B, same 224 px tower + linear projector
    "llava_plus_v0_13b": SynthConfig("llava_plus_v0_13b", 5120, 13824, 40, 40, 40, 32000, init="hf", v_image_size=224, mm_projector_type="linear"),
    "llava_plus_v0_7b": SynthConfig("llava_plus_v0_7b", 4096, 11008, 32, 32, 32, 32000, init="hf", v_image_size=224, mm_projector_type="linear"),
}


def with_layers(cfg: SynthConfig, n_dec: int, n_vis: int | None = None) -> SynthConfig:
    """Same widths, fewer layers (bounded CPU samples / mid-size parity at real geometry)."""
    return replace(cfg, name=f"{cfg.name}_L{n_dec}", num_hidden_layers=n_dec,
                   v_num_hidden_layers=cfg.v_num_hidden_layers if n_vis is None else n_vis)


def tensor_shapes(cfg: SynthConfig) -> Dict[str, tuple]:
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    D = cfg.head_dim
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    s: Dict[str, tuple] = {"model.embed_tokens.weight": (V, H), "model.norm.weight": (H,), "lm_head.weight": (V, H)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (nh * D, H)
        s[p + "self_attn.k_proj.weight"] = (nkv * D, H)
        s[p + "self_attn.v_proj.weight"] = (nkv * D, H)
        s[p + "self_attn.o_proj.weight"] = (H, nh * D)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
    Dv, Fv, ps = cfg.v_hidden_size, cfg.v_intermediate_size, cfg.v_patch_size
    s["vision.embeddings.class_embedding"] = (Dv,)
    s["vision.embeddings.patch_embedding.weight"] = (Dv, 3, ps, ps)
    s["vision.embeddings.position_embedding.weight"] = (cfg.num_patches + 1, Dv)
    s["vision.pre_layrnorm.weight"] = (Dv,)
    s["vision.pre_layrnorm.bias"] = (Dv,)
    for i in range(cfg.v_num_hidden_layers):
        p = f"vision.encoder.layers.{i}."
        for lin, shp in (("self_attn.q_proj", (Dv, Dv)), ("self_attn.k_proj", (Dv, Dv)), ("self_attn.v_proj", (Dv, Dv)),
                         ("self_attn.out_proj", (Dv, Dv)), ("mlp.fc1", (Fv, Dv)), ("mlp.fc2", (Dv, Fv))):
            s[p + lin + ".weight"] = shp
            s[p + lin + ".bias"] = (shp[0],)
        for ln in ("layer_norm1", "layer_norm2"):
            s[p + ln + ".weight"] = (Dv,)
            s[p + ln + ".bias"] = (Dv,)
    depth = cfg.projector_depth
    if cfg.mm_projector_type == "linear":
        s["mm_projector.weight"] = (H, Dv)
        s["mm_projector.bias"] = (H,)
    else:
        for j in range(depth):
            s[f"mm_projector.{2 * j}.weight"] = (H, Dv if j == 0 else H)
            s[f"mm_projector.{2 * j}.bias"] = (H,)
    return s


def make_tensor(cfg: SynthConfig, name: str, shape: tuple, seed: int) -> np.ndarray:
    """One tensor, generated independently of all others (so 7B-scale weights can be streamed tensor by tensor)."""
    import zlib
    rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    is_norm_w = name.endswith("layernorm.weight") or name.endswith("norm.weight") or "layer_norm" in name and name.endswith(".weight") \
        or name.endswith("pre_layrnorm.weight")
    if is_norm_w:
        if cfg.init == "hf":
            return np.ones(shape, np.float32)
        return (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
    if name.endswith(".bias"):
        if cfg.init == "hf":
            return np.zeros(shape, np.float32)
        return (0.05 * rs.standard_normal(shape)).astype(np.float32)
    if cfg.init == "hf":
        return (0.02 * rs.standard_normal(shape)).astype(np.float32)
    if name == "model.embed_tokens.weight" or name.startswith("vision.embeddings.class") or name.startswith("vision.embeddings.position"):
        return rs.standard_normal(shape).astype(np.float32) * (1.0 if name == "model.embed_tokens.weight" else 0.5)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return (rs.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def make_weights(cfg: SynthConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    return {n: make_tensor(cfg, n, shp, seed) for n, shp in tensor_shapes(cfg).items()}


def make_pixels(cfg: SynthConfig, n_images: int, seed: int = 1) -> np.ndarray:
    """`torch.rand`-like image in [0,1], CLIP-normalised -> pixel_values [N,3,S,S] (BASELINE.md §2 inputs)."""
    rs = np.random.RandomState(seed)
    img = rs.random_sample((n_images, 3, cfg.v_image_size, cfg.v_image_size)).astype(np.float32)
    mean = np.asarray(CLIP_MEAN, np.float32)[None, :, None, None]
    std = np.asarray(CLIP_STD, np.float32)[None, :, None, None]
    return ((img - mean) / std).astype(np.float32)


def make_prompt(cfg: SynthConfig, length: int, image_positions=(35,), seed: int = 2) -> np.ndarray:
    """input_ids [length]: ids[0] = 1 (BOS), IMAGE_TOKEN_INDEX at `image_positions`, the rest uniform in [3, vocab)."""
    rs = np.random.RandomState(seed)
    ids = rs.randint(3, cfg.vocab_size, size=(length,)).astype(np.int64)
    ids[0] = 1
    for p in image_positions:
        if p < length:
            ids[p] = IMAGE_TOKEN_INDEX
    return ids
