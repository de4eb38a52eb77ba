"""Import the REAL reference code from /root/reference (build container only) — TEST INFRASTRUCTURE.

Used by oracle/make_golden.py to pin the oracle: the reference's own llava_arch.py / llava_llama.py /
clip_encoder.py / multimodal_projector/builder.py run on CPU over transformers 5.15 with two compatibility shims
(SURVEY §8c):
  1. stub package objects for `llava`, `llava.model`, `llava.model.language_model` (their __path__ points into the
     reference tree) so `llava/__init__.py` -> MPT -> ImportError is skipped, plus `register(..., exist_ok=True)`
     because transformers now ships its own "llava" model type (llava_llama.py:110-111 would raise);
  2. a DynamicCache subclass with __getitem__ for the decode branch's `past_key_values[-1][-1].shape[-2]`
     (llava_arch.py:105).
/root/reference does not exist on the GPU box.  There the same files are imported SOURCELESS from oracle/_ref/llava_pyc/ (byte code compiled from
/root/reference by oracle/build_ref_worker.py in the build container; git-ignored, travels with the snapshot) — by bench.py's `cpu_baseline` leg only
(kind "reference": the reference's own forward + greedy loop timed on the host cores).  Never imported by the product package.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types
from typing import Dict

import numpy as np

_PYC_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "llava_pyc")
REF_ROOT = os.environ.get("LLAVA_REFERENCE_ROOT", "/root/reference")
if not os.path.isdir(os.path.join(REF_ROOT, "llava", "model")) and os.path.isfile(os.path.join(_PYC_ROOT, "llava", "model", "llava_arch.pyc")):
    REF_ROOT = _PYC_ROOT                       # GPU box: the byte-compiled copy of the same files


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "llava", "model"))


def is_sourceless() -> bool:
    return REF_ROOT == _PYC_ROOT


_loaded = None


def load_reference():
    """Returns the reference module namespace: LlavaLlamaForCausalLM, LlavaConfig, mm_utils, constants."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    import transformers
    from transformers import AutoConfig, AutoModelForCausalLM

    def stub(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_ROOT, rel)]
        m.__package__ = name
        sys.modules[name] = m
        return m

    for name, rel in (("llava", "llava"), ("llava.model", "llava/model"), ("llava.model.language_model", "llava/model/language_model")):
        if name not in sys.modules:
            stub(name, rel)

    orig_cfg_register = AutoConfig.register
    orig_model_register = AutoModelForCausalLM.register

    def cfg_register(model_type, config, exist_ok=False):
        return orig_cfg_register(model_type, config, exist_ok=True)

    def model_register(config_class, model_class, exist_ok=False):
        return orig_model_register(config_class, model_class, exist_ok=True)

    AutoConfig.register = staticmethod(cfg_register)
    AutoModelForCausalLM.register = classmethod(lambda cls, c, m, exist_ok=False: orig_model_register.__func__(cls, c, m, exist_ok=True))
    try:
        import llava.model.language_model.llava_llama as ll    # the reference's own file
        import llava.mm_utils as mm_utils
        import llava.constants as constants
    finally:
        AutoConfig.register = orig_cfg_register
        AutoModelForCausalLM.register = orig_model_register
    _loaded = types.SimpleNamespace(LlavaLlamaForCausalLM=ll.LlavaLlamaForCausalLM, LlavaConfig=ll.LlavaConfig, mm_utils=mm_utils,
                                    constants=constants, transformers_version=transformers.__version__)
    return _loaded


_loaded_train = None


def load_reference_train():
    """The reference's training-data side, imported from its own files: namespace(conversation = llava/conversation.py, train = llava/train/train.py).
    Two imports of those files do not resolve in this image and are stood in for while they load: torchvision (conversation.py uses it for a thumbnail
    resize in the web UI only) and llava/train/llava_trainer.py (an HF Trainer subclass against transformers 4.31 internals; train.py only names the class
    inside train()).  Build container only (needs the source tree)."""
    global _loaded_train
    if _loaded_train is not None:
        return _loaded_train
    load_reference()
    tv = [n for n in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional") if n not in sys.modules]
    for n in tv:
        sys.modules[n] = types.ModuleType(n)
    try:
        import llava.conversation as conversation
    finally:
        for n in tv:
            sys.modules.pop(n, None)
    if "llava.train" not in sys.modules:
        m = types.ModuleType("llava.train")
        m.__path__ = [os.path.join(REF_ROOT, "llava", "train")]
        m.__package__ = "llava.train"
        sys.modules["llava.train"] = m
    if "llava.train.llava_trainer" not in sys.modules:
        t = types.ModuleType("llava.train.llava_trainer")
        t.LLaVATrainer = type("LLaVATrainer", (object,), {})
        sys.modules["llava.train.llava_trainer"] = t
    import llava.train.train as train
    _loaded_train = types.SimpleNamespace(conversation=conversation, train=train)
    return _loaded_train


def subscriptable_cache():
    from transformers import DynamicCache

    class SubscriptableCache(DynamicCache):
        def __getitem__(self, i):
            layer = self.layers[i]
            return (layer.keys, layer.values)

    return SubscriptableCache()


def _clip_config(cfg):
    from transformers import CLIPVisionConfig
    return CLIPVisionConfig(hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size,
                            num_hidden_layers=cfg.v_num_hidden_layers, num_attention_heads=cfg.v_num_attention_heads,
                            image_size=cfg.v_image_size, patch_size=cfg.v_patch_size, layer_norm_eps=cfg.v_layer_norm_eps,
                            hidden_act="quick_gelu", projection_dim=cfg.v_hidden_size)


def build_reference_model(cfg, weights, dtype=None, fast_init: bool = False):
    """Construct the reference's LlavaLlamaForCausalLM (CPU, eager attention) holding exactly `weights` (numpy arrays or torch tensors by oracle name).
    dtype None: fp32 (the parity dtype of the goldens); fast_init: skip the random initialisation of the 7B-sized modules (every tensor is overwritten)."""
    import contextlib
    import torch
    from transformers import CLIPImageProcessor, CLIPVisionModel
    ref = load_reference()
    tmp = tempfile.mkdtemp(prefix="lmx_clip_")
    vcfg = _clip_config(cfg)
    clip = CLIPVisionModel(vcfg)
    sd = {}
    for k, v in weights.items():
        if k.startswith("vision."):
            sd["vision_model." + k[len("vision."):]] = _as_tensor(v).float()
    # tensors the tower owns but the path never reads (post_layernorm feeds only the pooled output)
    have = set(clip.state_dict().keys())
    extra = {k: clip.state_dict()[k] for k in have - set(sd.keys())}
    missing_ok = all("post_layernorm" in k or "position_ids" in k for k in extra)
    if not missing_ok:
        # transformers 5.x drops the `vision_model.` prefix; retry with bare names
        sd = {k[len("vision_model."):]: v for k, v in sd.items()}
        extra = {k: clip.state_dict()[k] for k in have - set(sd.keys())}
        assert all("post_layernorm" in k or "position_ids" in k for k in extra), sorted(extra)[:5]
    sd.update(extra)
    clip.load_state_dict(sd, strict=True)
    clip.save_pretrained(tmp)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image_size}, crop_size={"height": cfg.v_image_size, "width": cfg.v_image_size}).save_pretrained(tmp)

    lcfg = ref.LlavaConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                           num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                           vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                           max_position_embeddings=cfg.max_position_embeddings, hidden_act="silu", attention_bias=False,
                           mlp_bias=False, tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    lcfg.mm_vision_tower = tmp
    lcfg.mm_projector_type = cfg.mm_projector_type
    lcfg.mm_hidden_size = cfg.v_hidden_size
    lcfg.mm_vision_select_layer = cfg.mm_vision_select_layer
    lcfg.mm_vision_select_feature = cfg.mm_vision_select_feature
    lcfg.tokenizer_padding_side = cfg.tokenizer_padding_side
    lcfg.tokenizer_model_max_length = cfg.tokenizer_model_max_length
    lcfg.pretraining_tp = 1
    lcfg._attn_implementation = "eager"
    ctx = contextlib.nullcontext()
    if fast_init:
        from transformers.initialization import no_init_weights
        ctx = no_init_weights()
    old_default = torch.get_default_dtype()
    try:
        if dtype is not None:
            torch.set_default_dtype(dtype)              # allocate the decoder in the target dtype (13.5 GB instead of 27 GB at 7B)
        with ctx:
            model = ref.LlavaLlamaForCausalLM(lcfg)
    finally:
        torch.set_default_dtype(old_default)
    model.eval()
    tower = model.get_vision_tower()
    tower.load_model()                       # clip_encoder.py:21-27
    msd = {}
    for k, v in weights.items():
        if k.startswith("vision."):
            continue
        if k.startswith("mm_projector."):
            msd["model." + k] = _as_tensor(v)
        else:
            msd[k] = _as_tensor(v)
    own = model.state_dict()
    for k in own:
        if k.startswith("model.vision_tower."):
            msd[k] = own[k]
    missing = set(own) - set(msd)
    assert not missing, sorted(missing)[:5]
    model.load_state_dict(msd, strict=True)
    model.float() if dtype is None else model.to(dtype)
    return model


def _as_tensor(v):
    import torch
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
