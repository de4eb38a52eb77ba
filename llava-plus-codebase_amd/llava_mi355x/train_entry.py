"""Training entry of visual instruction tuning on the MI355X training step (BASELINE config 5): what llava/train/train.py:805-1000 does around
the HF Trainer, around llava_mi355x/train.py: TrainStep instead.

    initialize_vision_modules      llava/model/llava_arch.py:42-82     config fields of the vision side, a fresh mm_projector (torch.nn.Linear's own initialiser,
                                                                       drawn in the reference's order) or the rows of `pretrain_mm_mlp_adapter`
    smart_tokenizer_and_embedding_resize   llava/train/train.py:229-251   a [PAD] token for `--version v0` tokenizers without one
    initialize_vision_tokenizer    llava/model/llava_arch.py:242-284   <im_patch> / <im_start> / <im_end> tokens, embedding + lm_head rows for them (mean of the old rows),
                                                                       rows from the adapter file, which of the two matrices trains
    lr_at                          HF get_scheduler("cosine" | "linear" | "constant", warm-up = ceil(ratio x steps)) — the scripts' --lr_scheduler_type / --warmup_ratio
    save_checkpoint                llava/train/train.py:185-214        projector-only `mm_projector.bin` when tune_mm_mlp_adapter, else the whole state dict
    train                          llava/train/train.py:805-1000       template by --version, pad = unk, vision setup, dataset + collator (train_data.py), step loop

Here a model is a (config, HF-named weight dict) pair — TrainStep owns the parameters in its flat ZeRO-2 buffers, the CLIP tower is the frozen inference tower of
llava_mi355x/model.py (clip_encoder.py:25: `requires_grad_(False)`), its output enters the step as data.  Freezing (tune_mm_mlp_adapter, freeze_backbone,
freeze_mm_mlp_adapter) = the frozen tensors' gradients never reach the optimiser (TrainStep.set_trainable).  Out of scope, as SURVEY §2 has it: DeepSpeed / HF
Trainer plumbing, bitsandbytes, LoRA, MPT, checkpoint resumption."""
from __future__ import annotations

import json
import math
import os
from typing import Callable, Dict, Mapping, MutableMapping, Optional

import torch

from . import conversation as conversation_lib
from .constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN


def _proj_prefix(weights: Mapping[str, torch.Tensor]) -> str:
    return "mm_projector." if any(k.startswith("mm_projector.") for k in weights) else "model.mm_projector."


def projector_parameter_names(projector_type: str, prefix: str = "model.mm_projector."):
    """HF names of the projector's tensors (llava/model/multimodal_projector/builder.py:33-51): `linear` -> weight / bias, `mlpNx_gelu` -> Sequential slots
    0, 2, .. (GELU in between), `identity` -> none."""
    import re
    if projector_type == "linear":
        return [(prefix + "weight", prefix + "bias")]
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return [(f"{prefix}{2 * j}.weight", f"{prefix}{2 * j}.bias") for j in range(int(m.group(1)))]
    if projector_type == "identity":
        return []
    raise ValueError(f"Unknown projector type: {projector_type}")


def initialize_vision_modules(config, weights: MutableMapping[str, torch.Tensor], model_args, vision_hidden_size: int, dtype=torch.float32) -> None:
    """Vision side of a language model that is about to be tuned: config fields, a projector if the weights hold none, the adapter file's rows if one is named.
    A fresh projector is drawn with torch.nn.Linear's initialiser in the order the reference builds its layers (same torch seed -> same tensors)."""
    config.mm_vision_tower = model_args.vision_tower
    config.use_mm_proj = True
    config.mm_projector_type = getattr(model_args, "mm_projector_type", "linear")
    config.mm_hidden_size = int(vision_hidden_size)
    config.mm_vision_select_layer = model_args.mm_vision_select_layer
    config.mm_vision_select_feature = model_args.mm_vision_select_feature
    prefix = _proj_prefix(weights)
    names = projector_parameter_names(config.mm_projector_type, prefix)
    if names and names[0][0] not in weights:
        fan_in = config.mm_hidden_size
        for wn, bn in names:
            lin = torch.nn.Linear(fan_in, config.hidden_size)
            weights[wn], weights[bn] = lin.weight.detach().to(dtype), lin.bias.detach().to(dtype)
            fan_in = config.hidden_size
    path = getattr(model_args, "pretrain_mm_mlp_adapter", None)
    if path is not None:
        saved = torch.load(path, map_location="cpu")
        rows = {k.split("mm_projector.")[1]: v for k, v in saved.items() if "mm_projector" in k}
        want = {n[len(prefix):] for pair in names for n in pair}
        if set(rows) != want:
            raise RuntimeError(f"Error(s) in loading state_dict for mm_projector: file holds {sorted(rows)}, the projector needs {sorted(want)}")
        for k, v in rows.items():
            weights[prefix + k] = v.to(weights[prefix + k].dtype) if prefix + k in weights else v.to(dtype)


def _resize_token_matrix(w: torch.Tensor, n: int, std: float) -> torch.Tensor:
    """transformers 4.31's resize_token_embeddings: a new [n, dim] matrix drawn N(0, initializer_range) as a whole, the old rows copied over its first rows."""
    if n == w.shape[0]:
        return w
    new = torch.empty((n, w.shape[1]), dtype=torch.float32).normal_(mean=0.0, std=std).to(w.dtype)
    k = min(n, w.shape[0])
    new[:k] = w[:k]
    return new


def smart_tokenizer_and_embedding_resize(special_tokens_dict: Dict, tokenizer, config, weights: MutableMapping[str, torch.Tensor]) -> int:
    """`--version v0` with a tokenizer that has no pad token (llava/train/train.py:229-251, called at :886-892 with {"pad_token": "[PAD]"}): the special tokens are
    added, both token matrices grow to len(tokenizer), the new rows start as the mean of the old ones.  Returns the number of tokens added."""
    n_new = tokenizer.add_special_tokens(special_tokens_dict)
    std = float(getattr(config, "initializer_range", 0.02))
    for key in ("model.embed_tokens.weight", "lm_head.weight"):
        w = _resize_token_matrix(weights[key], len(tokenizer), std)
        if n_new > 0:
            w[-n_new:] = w[:-n_new].float().mean(dim=0, keepdim=True).to(w.dtype)
        weights[key] = w
    config.vocab_size = len(tokenizer)
    return n_new


def initialize_vision_tokenizer(config, weights: MutableMapping[str, torch.Tensor], model_args, tokenizer) -> Dict[str, object]:
    """Image tokens of the two optional prompt formats.  Returns {"num_new_tokens", "embed_tokens_trainable", "lm_head_trainable"}: the last two are None where
    the reference leaves requires_grad as it finds it."""
    emb, head = "model.embed_tokens.weight", "lm_head.weight"
    std = float(getattr(config, "initializer_range", 0.02))
    out: Dict[str, object] = {"num_new_tokens": 0, "embed_tokens_trainable": None, "lm_head_trainable": None}

    def resize():
        n = len(tokenizer)
        weights[emb] = _resize_token_matrix(weights[emb], n, std)
        weights[head] = _resize_token_matrix(weights[head], n, std)
        config.vocab_size = n

    if model_args.mm_use_im_patch_token:
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
        resize()
    if model_args.mm_use_im_start_end:
        n_new = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
        resize()
        out["num_new_tokens"] = n_new
        if n_new > 0:
            for key in (emb, head):
                w = weights[key]
                w[-n_new:] = w[:-n_new].float().mean(dim=0, keepdim=True).to(w.dtype)
        if model_args.tune_mm_mlp_adapter:
            out["embed_tokens_trainable"], out["lm_head_trainable"] = True, False
        if model_args.pretrain_mm_mlp_adapter:
            saved = torch.load(model_args.pretrain_mm_mlp_adapter, map_location="cpu")["model.embed_tokens.weight"]
            assert n_new == 2
            w = weights[emb]
            if w.shape == saved.shape:
                w[-n_new:] = saved[-n_new:].to(w.dtype)
            elif saved.shape[0] == n_new:
                w[-n_new:] = saved.to(w.dtype)
            else:
                raise ValueError(f"Unexpected embed_tokens_weight shape. Pretrained: {saved.shape}. Current: {w.shape}. Numer of new tokens: {n_new}.")
    elif model_args.mm_use_im_patch_token and model_args.tune_mm_mlp_adapter:
        out["embed_tokens_trainable"], out["lm_head_trainable"] = False, False
    return out


def lr_at(step: int, total_steps: int, base_lr: float, warmup_ratio: float = 0.0, kind: str = "cosine") -> float:
    """Learning rate of optimiser step `step` (0-based) as HF's get_scheduler gives it: linear warm-up over ceil(total x ratio) steps, then cosine to 0 /
    linear to 0 / constant."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    if kind in ("constant", "constant_with_warmup"):
        return base_lr
    progress = (step - warm) / max(1, total_steps - warm)
    if kind == "linear":
        return base_lr * max(0.0, 1.0 - progress)
    if kind == "cosine":
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
    raise ValueError(f"unsupported lr_scheduler_type {kind!r}")


def save_checkpoint(output_dir: str, config, state: Mapping[str, torch.Tensor], model_args) -> str:
    """tune_mm_mlp_adapter: only what that stage trains — `model.mm_projector.*` (+ `model.embed_tokens.weight` with mm_use_im_start_end) — as
    `mm_projector.bin`, the file `pretrain_mm_mlp_adapter` reads back.  Otherwise the whole state dict (HF names) + config.json."""
    os.makedirs(output_dir, exist_ok=True)
    named = {(("model." + k) if k.startswith("mm_projector.") else k): v.detach().cpu() for k, v in state.items()}
    if getattr(model_args, "tune_mm_mlp_adapter", False):
        keys = ["mm_projector"] + (["embed_tokens", "embed_in"] if getattr(model_args, "mm_use_im_start_end", False) else [])
        path = os.path.join(output_dir, "mm_projector.bin")
        torch.save({k: v for k, v in named.items() if any(m in k for m in keys)}, path)
    else:
        path = os.path.join(output_dir, "pytorch_model.bin")
        torch.save(named, path)
    if hasattr(config, "to_json_string"):
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            f.write(config.to_json_string())
    return path


def train(model_args, data_args, training_args, *, config, weights: MutableMapping[str, torch.Tensor], tokenizer, vision_tower: Optional[Callable] = None,
          group=None, device="cuda", log: Optional[Callable[[dict], None]] = None) -> Dict[str, object]:
    """The reference's train() with the loaded pieces passed in: `config` + `weights` = the language model (HF names; builder.iter_checkpoint reads a
    checkpoint into such a dict), `tokenizer`, `vision_tower` = the frozen tower of the inference model (LlavaLlamaForCausalLM.get_vision_tower(): callable
    on pixel batches, with .image_processor and .hidden_size).  One process per GPU; `group` = the data-parallel group (ZeRO-2 inside TrainStep; every rank
    walks its own stride of the shuffled sample order).  Returns {"losses", "steps", "checkpoint", "state"}."""
    from .train import TrainStep
    from .train_data import make_supervised_data_module
    dtype = torch.float16 if training_args.fp16 else (torch.bfloat16 if training_args.bf16 else torch.float32)
    assert training_args.gradient_accumulation_steps == 1, "gradient accumulation is not wired (TrainStep holds one micro-batch of gradients)"
    # ---- prompt template, padding (train.py:885-907) ----------------------------------------------------------------------------------------------------
    if model_args.version == "v0":
        if tokenizer.pad_token is None:
            smart_tokenizer_and_embedding_resize(dict(pad_token="[PAD]"), tokenizer, config, weights)
    else:
        tokenizer.pad_token = tokenizer.unk_token
        if model_args.version != "v0.5":
            conversation_lib.default_conversation = conversation_lib.conv_templates.get(model_args.version, conversation_lib.conv_templates["vicuna_v1"])
    # ---- vision side (train.py:909-948) ---------------------------------------------------------------------------------------------------------------------
    frozen = set()
    if model_args.vision_tower is not None:
        if vision_tower is None:
            raise ValueError("model_args.vision_tower is set: pass the frozen tower (LlavaLlamaForCausalLM.get_vision_tower()) as vision_tower=")
        initialize_vision_modules(config, weights, model_args, vision_tower.hidden_size, dtype=dtype)
        data_args.image_processor = vision_tower.image_processor
        data_args.is_multimodal = True
        config.image_aspect_ratio = data_args.image_aspect_ratio
        config.tokenizer_padding_side = tokenizer.padding_side
        config.tokenizer_model_max_length = tokenizer.model_max_length
        config.tune_mm_mlp_adapter = training_args.tune_mm_mlp_adapter = model_args.tune_mm_mlp_adapter
        config.freeze_mm_mlp_adapter = training_args.freeze_mm_mlp_adapter
        config.mm_use_im_start_end = data_args.mm_use_im_start_end = model_args.mm_use_im_start_end
        config.mm_projector_lr = training_args.mm_projector_lr
        config.mm_use_im_patch_token = model_args.mm_use_im_patch_token
        tok_state = initialize_vision_tokenizer(config, weights, model_args, tokenizer)
        is_proj = lambda k: "mm_projector" in k
        if model_args.freeze_backbone:
            frozen |= {k for k in weights if k.startswith("model.") and not is_proj(k)}
        if model_args.tune_mm_mlp_adapter:
            frozen |= {k for k in weights if not is_proj(k)}
        if training_args.freeze_mm_mlp_adapter:
            frozen |= {k for k in weights if is_proj(k)}
        for key, flag in (("model.embed_tokens.weight", tok_state["embed_tokens_trainable"]), ("lm_head.weight", tok_state["lm_head_trainable"])):
            if flag is True:
                frozen.discard(key)
            elif flag is False:
                frozen.add(key)
    elif model_args.freeze_backbone:
        frozen |= {k for k in weights if k.startswith("model.")}
    if training_args.mm_projector_lr is not None:
        raise NotImplementedError("mm_projector_lr (a second learning rate for the projector) is not wired")
    # ---- data (train.py:963-964) ------------------------------------------------------------------------------------------------------------------------------
    module = make_supervised_data_module(tokenizer=tokenizer, data_args=data_args)
    dataset, collate = module["train_dataset"], module["data_collator"]
    world, rank = 1, 0
    if group is not None:
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = int(training_args.per_device_train_batch_size)
    per_epoch = len(dataset) // (B * world)
    if per_epoch == 0:
        raise ValueError(f"{len(dataset)} samples do not fill one batch of {B} x {world}")
    total = training_args.max_steps if training_args.max_steps > 0 else int(math.ceil(training_args.num_train_epochs * per_epoch))
    # ---- the step ----------------------------------------------------------------------------------------------------------------------------------------------
    ts = TrainStep(config, {k: v for k, v in weights.items() if "vision_tower" not in k and not k.startswith("vision.")}, dtype=dtype, device=device,
                   lr=training_args.learning_rate, betas=(training_args.adam_beta1, training_args.adam_beta2), eps=training_args.adam_epsilon,
                   weight_decay=training_args.weight_decay, max_grad_norm=training_args.max_grad_norm, group=group,
                   checkpoint=bool(training_args.gradient_checkpointing), max_positions=max(2048, int(tokenizer.model_max_length) + 1024))
    ts.set_trainable(lambda name: (("model." + name) if name.startswith("mm_projector.") else name) not in frozen)
    gen = torch.Generator().manual_seed(int(training_args.seed))
    sampler = None
    if training_args.group_by_modality_length:
        # LLaVATrainer._get_train_sampler (llava/train/llava_trainer.py:144-158): batches of similar length, one modality per megabatch
        from .train_data import LengthGroupedSampler
        lengths = [l for part in dataset.datasets for l in part.modality_lengths]
        sampler = LengthGroupedSampler(B, world_size=world, lengths=lengths, generator=gen, group_by_modality=True)
    losses, step = [], 0
    while step < total:
        # the same order on every rank (same seed): rank r takes batch r of every group of `world`
        order = list(iter(sampler)) if sampler is not None else torch.randperm(len(dataset), generator=gen).tolist()
        for b in range(per_epoch):
            if step >= total:
                break
            first = (b * world + rank) * B
            batch = collate([dataset[i] for i in order[first: first + B]])
            feats = None
            if "images" in batch:
                pix = batch["images"]
                if isinstance(pix, (list, tuple)):
                    raise NotImplementedError("images of different shapes in one batch")
                feats = vision_tower(pix.to(device=device, dtype=dtype))
            ts.lr = lr_at(step, total, training_args.learning_rate, training_args.warmup_ratio, training_args.lr_scheduler_type)
            loss, count = ts.step(batch["input_ids"], batch["labels"], batch["attention_mask"], image_features=feats)
            step += 1
            if training_args.logging_steps and step % training_args.logging_steps == 0:
                rec = {"step": step, "loss": float(loss.item()), "learning_rate": ts.lr, "label_positions": int(count)}
                losses.append(rec["loss"])
                if log is not None:
                    log(rec)
    state = ts.state_dict()
    path = save_checkpoint(training_args.output_dir, config, state, model_args) if rank == 0 and training_args.output_dir else None
    if rank == 0 and training_args.output_dir:
        with open(os.path.join(training_args.output_dir, "trainer_state.json"), "w") as f:
            json.dump({"global_step": step, "log_history": [{"step": (i + 1) * max(1, int(training_args.logging_steps or 1)), "loss": l} for i, l in enumerate(losses)]}, f)   # a record every logging_steps steps
    return {"losses": losses, "steps": step, "checkpoint": path, "state": state, "frozen": sorted(frozen)}
