"""load_pretrained_model for the MI355X engine — same signature and 4-tuple as the reference's
llava/model/builder.py:26-151: (tokenizer, model, image_processor, context_len).

Supported: full LLaVA checkpoints in HF layout (`*.safetensors` or `pytorch_model*.bin` shards, key names as saved by
the reference incl. `model.mm_projector.*` and optionally `model.vision_tower.*`), projector-only checkpoints on a
base LLM (`mm_projector.bin`, builder.py:82-99), un-merged LoRA checkpoints on a base LLM (peft's `adapter_config.json` +
`adapter_model.*` + `non_lora_trainables.bin`, builder.py:50-81: merged into the base weights while they are loaded — peft
itself is not needed), the CLIP tower from `config.mm_vision_tower` (both the 4.31 `vision_model.*` and the 5.x bare key
layouts).  Out of scope, raise NotImplementedError: 8-bit/4-bit (bitsandbytes), MPT.
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


def read_lora_adapter(model_path: str):
    """peft's `save_pretrained` output as the reference's LoRA training writes it (llava/train/train.py:861-878 -> adapter_config.json,
    adapter_model.safetensors | adapter_model.bin).  Returns ({'<module path>.weight': (A [r, in], B [out, r])}, scaling) with module
    paths in checkpoint naming (`model.layers.0.self_attn.q_proj`), i.e. the operands of peft's merge, W += (B @ A) * scaling
    (LoraLayer.get_delta_weight: scaling = lora_alpha / r, or lora_alpha / sqrt(r) with use_rslora)."""
    import json
    with open(os.path.join(model_path, "adapter_config.json")) as f:
        ac = json.load(f)
    if ac.get("peft_type", "LORA") != "LORA":
        raise NotImplementedError(f"adapter type {ac.get('peft_type')}: only LoRA adapters are merged")
    if ac.get("bias", "none") != "none" or ac.get("fan_in_fan_out", False) or ac.get("use_dora", False) or ac.get("modules_to_save"):
        raise NotImplementedError("LoRA adapters with trained biases, fan_in_fan_out, DoRA or modules_to_save are not supported")
    r, alpha = int(ac["r"]), float(ac["lora_alpha"])
    scaling = alpha / (r ** 0.5) if ac.get("use_rslora", False) else alpha / r
    files = [os.path.join(model_path, n) for n in ("adapter_model.safetensors", "adapter_model.bin") if os.path.isfile(os.path.join(model_path, n))]
    if not files:
        raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {model_path}")
    halves = {}
    for k, v in iter_checkpoint(files[0]):
        for tag in (".lora_A", ".lora_B"):
            i = k.find(tag)
            if i < 0:
                continue
            mod = k[:i]
            while mod.startswith("base_model.model."):                      # PeftModel -> LoraModel -> the wrapped LlavaLlamaForCausalLM
                mod = mod[len("base_model.model."):]
            if mod.startswith("base_model."):
                mod = mod[len("base_model."):]
            halves.setdefault(mod + ".weight", {})[tag] = v
    pairs = {}
    for name, h in halves.items():
        if ".lora_A" not in h or ".lora_B" not in h:
            raise ValueError(f"adapter has only one LoRA factor for {name}")
        A, B = h[".lora_A"], h[".lora_B"]
        if A.dim() != 2 or B.dim() != 2 or A.shape[0] != B.shape[1]:
            raise ValueError(f"LoRA factors of {name} do not multiply: A {tuple(A.shape)}, B {tuple(B.shape)}")
        pairs[name] = (A, B)
    if not pairs:
        raise ValueError(f"{files[0]} holds no lora_A / lora_B tensors")
    return pairs, scaling


def merge_lora(W: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scaling: float, device) -> torch.Tensor:
    """peft's LoRA merge for one nn.Linear weight, W [out, in] + (B [out, r] @ A [r, in]) * scaling, on the device in fp32 through the
    engine's own GEMM (C = X . W^T + residual with X = B, W = (scaling A)^T, residual = the base weight; exact fp32 accumulation), rounded
    once when the caller loads it in the model dtype.  (peft adds the delta in the weight's 16-bit dtype: two roundings.)"""
    from . import ops
    out_f, in_f = W.shape
    r = A.shape[0]
    if tuple(A.shape) != (r, in_f) or tuple(B.shape) != (out_f, r):
        raise ValueError(f"LoRA factors A {tuple(A.shape)} / B {tuple(B.shape)} do not fit the weight {tuple(W.shape)}")
    rp = (r + 15) // 16 * 16                                               # the fp32 GEMM walks K in steps of 16: zero columns add nothing
    x = torch.zeros((out_f, rp), dtype=torch.float32)
    x[:, :r] = B.float()
    w = torch.zeros((in_f, rp), dtype=torch.float32)
    w[:, :r] = (A.float() * scaling).t()
    return ops.gemm(x.to(device), w.to(device), residual=W.to(device=device, dtype=torch.float32).contiguous())


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
                    device_map="auto", tp_rank: int = 0, tp_world: int = 1, max_position: Optional[int] = None, vocab_headroom: int = 8,
                    overrides=None, lora=None, **_unused):
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
    # overrides: {checkpoint key: tensor} that replace (or add to) the checkpoint's tensors — `non_lora_trainables.bin` (builder.py:58-74);
    # lora: (pairs, scaling) from read_lora_adapter: every weight named there is loaded as W + (B @ A) * scaling (builder.py:76-80, peft merge_and_unload)
    overrides = dict(overrides or {})
    pairs, scaling = lora if lora is not None else ({}, 0.0)
    merged = set()

    def put(k, v):
        cname = model.canonical_name(k)
        if cname is None or cname.startswith("vision."):
            return
        if k in pairs:
            A, B = pairs[k]
            v = merge_lora(v, A, B, scaling, model.device)
            merged.add(k)
        model.load_tensor(cname, v)

    for k, v in iter_checkpoint(model_path):
        put(k, overrides.pop(k, v))
    for k, v in overrides.items():                                             # tensors the base checkpoint does not hold (the projector)
        put(k, v)
    missing = sorted(set(pairs) - merged)
    if missing:
        raise KeyError(f"LoRA adapter targets weights the base checkpoint does not hold: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
    return model


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto", device="cuda",
                          torch_dtype: Optional[torch.dtype] = None, tp_rank: int = 0, tp_world: int = 1, max_position: Optional[int] = None):
    """Same steps, in the same order, as llava/model/builder.py:26-151 for the LLaVA / LLaMA family."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8-bit/4-bit loading is CUDA-only and out of scope of the MI355X path")
    name_l = model_name.lower()
    if "mpt" in name_l:
        raise NotImplementedError("MPT checkpoints are not supported by the MI355X path (LLaMA/Vicuna family only)")
    if "llava" not in name_l:
        raise NotImplementedError("plain language-model checkpoints: use a llava checkpoint (model_name must contain 'llava')")
    from transformers import AutoTokenizer
    kw = dict(torch_dtype=torch_dtype, device=device, tp_rank=tp_rank, tp_world=tp_world, max_position=max_position)
    if "lora" in name_l and model_base is None:                  # builder.py:48-49: warn, then load model_path as a full checkpoint
        import warnings
        warnings.warn("There is `lora` in model name but no `model_base` is provided. If you are loading a LoRA model, please provide the `model_base` argument.")
    if "lora" in name_l and model_base is not None:              # un-merged LoRA checkpoint on a base LLM (builder.py:50-81)
        tokenizer = AutoTokenizer.from_pretrained(model_base, use_fast=False)
        extra = {}
        nlt = os.path.join(model_path, "non_lora_trainables.bin")
        if os.path.exists(nlt):                                  # (the reference falls back to the HF hub here: no network in this environment)
            extra = torch.load(nlt, map_location="cpu", weights_only=True)
            extra = {(k[11:] if k.startswith("base_model.") else k): v for k, v in extra.items()}          # builder.py:71-73, verbatim rules
            if any(k.startswith("model.model.") for k in extra):
                extra = {(k[6:] if k.startswith("model.") else k): v for k, v in extra.items()}
        model = from_pretrained(model_base, config=LlavaConfig.from_pretrained(model_path), overrides=extra, lora=read_lora_adapter(model_path), **kw)
    elif model_base is not None:                                 # projector-only checkpoint on a base LLM (builder.py:82-99)
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
    # reuse between the turns of a conversation (the LLaVA-Plus tool loop re-sends image + first exchange: gradio_web_server_llava_plus.py:600-637), opt-in:
    # LLAVA_MI355X_REUSE=1 or "<images>,<prefixes>" (reuse.py: image features by pixel content, KV prefixes of finished requests)
    ru = os.environ.get("LLAVA_MI355X_REUSE", "0") or "0"
    if ru != "0":
        parts = [int(v) for v in ru.split(",")] if "," in ru else []
        model.enable_reuse(images=parts[0] if parts else 64, prefixes=parts[1] if len(parts) > 1 else 32)
    from . import mm_utils
    mm_utils.set_device_preprocess_model(model)          # used by process_images when LLAVA_MI355X_DEVICE_PREPROCESS=1
    context_len = getattr(model.config, "max_sequence_length", 2048)     # builder.py:146-149
    return tokenizer, model, image_processor, context_len
