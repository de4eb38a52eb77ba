"""load_pretrained_model for the MI355X engine — same signature and 4-tuple as the reference's
llava/model/builder.py:26-151: (tokenizer, model, image_processor, context_len).

Supported: full LLaVA checkpoints in HF layout (`*.safetensors` or `pytorch_model*.bin` shards, key names as saved by
the reference incl. `model.mm_projector.*` and optionally `model.vision_tower.*`), projector-only checkpoints on a
base LLM (`mm_projector.bin`, builder.py:82-99), the CLIP tower from `config.mm_vision_tower` (both the 4.31
`vision_model.*` and the 5.x bare key layouts).  Out of scope, raise NotImplementedError: 8-bit/4-bit (bitsandbytes),
un-merged LoRA (peft), MPT.
"""
from __future__ import annotations

import glob
import os
from typing import Iterator, Optional, Tuple

import torch

from .constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
from .model import LlavaConfig, LlavaLlamaForCausalLM


def iter_checkpoint(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (key, tensor) from every weight shard in a HF-format directory (or a single file)."""
    files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    yield k, sf.get_tensor(k)
        else:
            try:
                sd = torch.load(f, map_location="cpu", mmap=True, weights_only=True)
            except (RuntimeError, ValueError):          # legacy (non-zip) .bin shards cannot be memory-mapped; the reference loads them too
                sd = torch.load(f, map_location="cpu", weights_only=True)
            for k, v in sd.items():
                yield k, v


def load_vision_tower(model: LlavaLlamaForCausalLM) -> None:
    """CLIPVisionTower.load_model (clip_encoder.py:21-27): image processor + tower weights into the engine."""
    from transformers import CLIPImageProcessor
    name = model.config.mm_vision_tower
    tower = model.get_vision_tower()
    if not os.path.isdir(name):
        raise FileNotFoundError(f"vision tower '{name}' is not a local directory (no network in this environment)")
    tower.image_processor = CLIPImageProcessor.from_pretrained(name)
    for k, v in iter_checkpoint(name):
        cname = model.canonical_name(k)
        if cname is None or not cname.startswith("vision."):
            continue
        model.load_tensor(cname, v)


def from_pretrained(model_path, config: Optional[LlavaConfig] = None, torch_dtype: Optional[torch.dtype] = None, device="cuda", low_cpu_mem_usage=True,
                    device_map="auto", tp_rank: int = 0, tp_world: int = 1, max_position: Optional[int] = None, vocab_headroom: int = 8, **_unused):
    """LlavaLlamaForCausalLM.from_pretrained (builder.py:100, 106): config + every language-model / projector tensor of the checkpoint.
    Like the reference (the tower is built with delay_load and loaded from `config.mm_vision_tower`, clip_encoder.py:15-27), tower
    tensors stored inside the LLaVA checkpoint are ignored."""
    from transformers import CLIPVisionConfig
    dtype = torch_dtype or {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[os.environ.get("LLAVA_MI355X_DTYPE", "bf16")]
    config = config if config is not None else LlavaConfig.from_pretrained(model_path)
    vcfg = CLIPVisionConfig.from_pretrained(config.mm_vision_tower)
    if not hasattr(config, "mm_vision_select_layer"):
        config.mm_vision_select_layer = -2
    model = LlavaLlamaForCausalLM(config, vcfg, dtype=dtype, device=device, tp_rank=tp_rank, tp_world=tp_world, max_position=max_position,
                                  vocab_headroom=vocab_headroom)
    for k, v in iter_checkpoint(model_path):
        cname = model.canonical_name(k)
        if cname is None or cname.startswith("vision."):
            continue
        model.load_tensor(cname, v)
    return model


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto", device="cuda",
                          torch_dtype: Optional[torch.dtype] = None, tp_rank: int = 0, tp_world: int = 1, max_position: Optional[int] = None):
    """Same steps, in the same order, as llava/model/builder.py:26-151 for the LLaVA / LLaMA family."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8-bit/4-bit loading is CUDA-only and out of scope of the MI355X path")
    name_l = model_name.lower()
    if "mpt" in name_l:
        raise NotImplementedError("MPT checkpoints are not supported by the MI355X path (LLaMA/Vicuna family only)")
    if "lora" in name_l:
        raise NotImplementedError("load LoRA checkpoints after merging them offline (scripts/merge_lora_weights.py in the reference)")
    if "llava" not in name_l:
        raise NotImplementedError("plain language-model checkpoints: use a llava checkpoint (model_name must contain 'llava')")
    from transformers import AutoTokenizer
    kw = dict(torch_dtype=torch_dtype, device=device, tp_rank=tp_rank, tp_world=tp_world, max_position=max_position)
    if model_base is not None:                                   # projector-only checkpoint on a base LLM (builder.py:82-99)
        tokenizer = AutoTokenizer.from_pretrained(model_base, use_fast=False)
        model = from_pretrained(model_base, config=LlavaConfig.from_pretrained(model_path), **kw)
        proj = torch.load(os.path.join(model_path, "mm_projector.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict({(k if k.startswith("model.") else "model." + k): v for k, v in proj.items()}, strict=False)
    else:
        tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
        model = from_pretrained(model_path, **kw)

    # builder.py:131-138
    if getattr(model.config, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if getattr(model.config, "mm_use_im_start_end", False):
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))

    # builder.py:140-144
    vision_tower = model.get_vision_tower()
    if not vision_tower.is_loaded:
        vision_tower.load_model()
    vision_tower.to(device=device, dtype=torch.float16)
    image_processor = vision_tower.image_processor
    model.finalize_weights()
    # continuous batching for model_worker's thread-per-request serving (llava/serve/model_worker.py:174-185): opt-in by env so the
    # reference's call signature stays untouched.  LLAVA_MI355X_BATCH=<capacity> (e.g. 32)
    cap = int(os.environ.get("LLAVA_MI355X_BATCH", "0") or 0)
    if cap > 1:
        # LLAVA_MI355X_PACKED_PREFILL=0: prefills stay on the request threads (one at a time) instead of being packed by the scheduler
        model.enable_batching(capacity=cap, packed_prefill=os.environ.get("LLAVA_MI355X_PACKED_PREFILL", "1") != "0")
    from . import mm_utils
    mm_utils.set_device_preprocess_model(model)          # used by process_images when LLAVA_MI355X_DEVICE_PREPROCESS=1
    context_len = getattr(model.config, "max_sequence_length", 2048)     # builder.py:146-149
    return tokenizer, model, image_processor, context_len
