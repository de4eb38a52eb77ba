"""Python mirror of the reference's model API over the MI355X engine (C ABI in include/llava_mi355x.h).

Same names / argument meaning / results / error behaviour as the reference classes, so `llava.serve.model_worker`,
`cli.py`, `predict.py` and the eval drivers keep working on top of it:
  LlavaConfig, LlavaLlamaModel, LlavaLlamaForCausalLM     llava/model/language_model/llava_llama.py:31-111
  LlavaMetaForCausalLM.encode_images / prepare_inputs_labels_for_multimodal   llava/model/llava_arch.py:94-240
  CLIPVisionTower (attributes callers touch)                                  llava/model/multimodal_encoder/clip_encoder.py:7-78
PyTorch tensors are containers (allocation, pointers, current stream); all arithmetic on the path runs in the HIP
kernels.  There is no CPU fallback: constructing a model without a GPU / without the built extension raises.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Sequence, Dict, List, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from . import _C
from ._C import LmxConfig, check, lib, ptr, stream_handle, torch_dtype_code
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX

try:  # HF containers for the return type / config plumbing only
    from transformers import LlamaConfig
    from transformers.modeling_outputs import CausalLMOutputWithPast
except Exception as _e:  # pragma: no cover
    raise ImportError("transformers is required for LlamaConfig / CausalLMOutputWithPast containers") from _e


class LlavaConfig(LlamaConfig):
    """llava_llama.py:31-32.  Extra fields read by the path: mm_vision_tower, mm_hidden_size, mm_projector_type,
    mm_vision_select_layer, mm_vision_select_feature, mm_use_im_start_end, mm_use_im_patch_token, image_aspect_ratio,
    tokenizer_padding_side, tokenizer_model_max_length (llava_arch.py:48-68, train.py:935-956)."""
    model_type = "llava"


_PROJ_RE = __import__("re").compile(r"^mlp(\d+)x_gelu$")


def _projector_kind(name: str) -> Tuple[int, int]:
    if name == "linear":
        return _C.PROJ_LINEAR, 1
    if name == "identity":
        return _C.PROJ_IDENTITY, 0
    m = _PROJ_RE.match(name)
    if m:
        return _C.PROJ_MLP_GELU, int(m.group(1))
    raise ValueError(f"Unknown projector type: {name}")          # multimodal_projector/builder.py:51


class LmxKVCache:
    """`past_key_values` of this build: one engine sequence (KV cache + device decode state) per batch row.
    Supports the one access the reference makes — `past_key_values[-1][-1].shape[-2]` (llava_arch.py:105)."""

    class _Shape:
        def __init__(self, shape):
            self.shape = shape

    def __init__(self, model: "LlavaLlamaForCausalLM", batch: int):
        self.model = model
        self.seqs: List[ctypes.c_void_p] = []
        for _ in range(batch):
            h = ctypes.c_void_p()
            check(lib.lmx_seq_create(model._h, ctypes.byref(h)), "lmx_seq_create")
            self.seqs.append(h)

    def lengths(self) -> List[int]:
        return [lib.lmx_seq_length(s) for s in self.seqs]

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return max(self.lengths()) if self.seqs else 0

    def __len__(self):
        return self.model.config.num_hidden_layers

    def __getitem__(self, i):
        c = self.model.config
        shp = (len(self.seqs), c.num_key_value_heads, self.get_seq_length(), c.hidden_size // c.num_attention_heads)
        return (self._Shape(shp), self._Shape(shp))

    def close(self):
        for s in self.seqs:
            lib.lmx_seq_destroy(s)
        self.seqs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CLIPVisionTower:
    """The attributes/methods callers touch on `model.get_vision_tower()` (builder.py:139-144, model_worker.py:148):
    is_loaded, load_model(), to(), image_processor, config, hidden_size, num_patches, dtype, device, __call__."""

    def __init__(self, owner: "LlavaLlamaForCausalLM"):
        self._owner = owner
        self.is_loaded = False
        self.image_processor = None
        self.vision_tower_name = getattr(owner.config, "mm_vision_tower", None)
        self.select_layer = owner.config.mm_vision_select_layer
        self.select_feature = getattr(owner.config, "mm_vision_select_feature", "patch")

    def load_model(self):
        """clip_encoder.py:21-27: processor + weights.  Weights go straight into the engine (see builder.load_vision_tower)."""
        from .builder import load_vision_tower
        load_vision_tower(self._owner)
        self.is_loaded = True

    def to(self, *args, **kwargs):
        return self

    def requires_grad_(self, flag=False):
        return self

    @property
    def config(self):
        return self._owner.vision_config

    @property
    def hidden_size(self):
        return self._owner.vision_config.hidden_size

    @property
    def num_patches(self):
        c = self._owner.vision_config
        return (c.image_size // c.patch_size) ** 2

    @property
    def dtype(self):
        return self._owner.dtype

    @property
    def device(self):
        return self._owner.device

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    def feature_rows(self) -> int:
        return self._owner.tokens_per_image

    @torch.no_grad()
    def forward(self, images):
        """clip_encoder.py:39-51: hidden_states[select_layer] (CLS dropped for 'patch'), cast back to the input dtype; a list of
        [3,S,S] images gives a list of [1,P,D] features.  (model.encode_images runs the same tower fused with the projector.)"""
        o = self._owner
        o._ensure_final()

        def run(x):
            x = x.to(device=o.device, dtype=o.dtype).contiguous()
            out = torch.empty((x.shape[0], o.tokens_per_image, self.hidden_size), dtype=o.dtype, device=o.device)
            with torch.cuda.device(o.device):
                check(lib.lmx_vision_tower(o._h, ptr(x), x.shape[0], ptr(out), stream_handle()), "lmx_vision_tower")
            return out

        if type(images) is list:
            return [run(im.unsqueeze(0)).to(im.dtype) for im in images]
        return run(images).to(images.dtype)

    __call__ = forward


class _EmbedTokens:
    def __init__(self, owner):
        self._owner = owner

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        o = self._owner
        flat = ids.reshape(-1).to(device=o.device, dtype=torch.int32).contiguous()
        if flat.numel() and bool(((flat < 0) | (flat >= o.config.vocab_size)).any()):
            raise IndexError("index out of range in self")          # what nn.Embedding raises for an id beyond the (resized) vocabulary
        out = torch.empty((flat.numel(), o.config.hidden_size), dtype=o.dtype, device=o.device)
        if flat.numel():
            check(lib.lmx_gather_embeds(o._h, ptr(flat), flat.numel(), None, ptr(out), stream_handle()), "gather_embeds")
        return out.view(*ids.shape, o.config.hidden_size)


class LlavaLlamaModel:
    """`model.get_model()` surface: get_vision_tower(), embed_tokens, mm_projector marker (llava_arch.py:27-40)."""

    def __init__(self, owner):
        self._owner = owner
        self.vision_tower = CLIPVisionTower(owner) if getattr(owner.config, "mm_vision_tower", None) is not None else None
        self.embed_tokens = _EmbedTokens(owner)
        self.mm_projector = getattr(owner.config, "mm_projector_type", "linear")

    def get_vision_tower(self):
        return self.vision_tower


class LlavaLlamaForCausalLM:
    """Drop-in for llava_llama.py:40-108 on the MI355X engine."""

    config_class = LlavaConfig

    def __init__(self, config: LlavaConfig, vision_config=None, dtype: torch.dtype = torch.bfloat16, device: Union[str, torch.device] = "cuda",
                 tp_rank: int = 0, tp_world: int = 1, max_position: Optional[int] = None, gemm_variant: int = 0, vocab_headroom: int = 0):
        """vocab_headroom: spare embedding / lm_head rows so that `resize_token_embeddings(len(tokenizer))` (builder.py:138) can grow the
        vocabulary by the <im_patch> / <im_start> / <im_end> tokens after the weights were loaded; the engine tables hold
        round_up(vocab_size + headroom, 8) rows, ids >= config.vocab_size are never picked and are sliced off the logits."""
        if not torch.cuda.is_available():
            raise RuntimeError("llava_mi355x needs an MI355X (HIP) device; there is no CPU path")
        self.config = config
        self.vision_config = vision_config
        self.dtype = dtype
        self.device = torch.device(device if str(device) != "cuda" else f"cuda:{torch.cuda.current_device()}")
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.generation_config = None
        c = LmxConfig()
        c.abi_version = _C.LMX_ABI_VERSION
        c.dtype = torch_dtype_code(dtype)
        c.hidden_size = config.hidden_size
        c.intermediate_size = config.intermediate_size
        c.n_layers = config.num_hidden_layers
        c.n_heads = config.num_attention_heads
        c.n_kv_heads = getattr(config, "num_key_value_heads", None) or config.num_attention_heads
        c.head_dim = config.hidden_size // config.num_attention_heads
        self._vocab_cap = (int(config.vocab_size) + int(vocab_headroom) + 7) // 8 * 8
        c.vocab_size = self._vocab_cap
        c.rms_eps = config.rms_norm_eps
        c.rope_theta = float(_rope_theta(config))
        c.max_position = int(max_position or config.max_position_embeddings)
        if vision_config is not None:
            c.v_hidden = vision_config.hidden_size
            c.v_intermediate = vision_config.intermediate_size
            c.v_layers = vision_config.num_hidden_layers
            c.v_heads = vision_config.num_attention_heads
            c.v_image_size = vision_config.image_size
            c.v_patch_size = vision_config.patch_size
            c.v_ln_eps = vision_config.layer_norm_eps
            c.select_layer = config.mm_vision_select_layer
            c.select_feature = _C.FEATURE_PATCH if getattr(config, "mm_vision_select_feature", "patch") == "patch" else _C.FEATURE_CLS_PATCH
            if getattr(config, "mm_vision_select_feature", "patch") not in ("patch", "cls_patch"):
                raise ValueError(f"Unexpected select feature: {config.mm_vision_select_feature}")   # clip_encoder.py:36
            c.projector_type, c.projector_depth = _projector_kind(getattr(config, "mm_projector_type", "linear"))
        c.tp_rank, c.tp_world = tp_rank, tp_world
        c.gemm_variant = gemm_variant
        self._cfg = c
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.lmx_create(ctypes.byref(c), ctypes.byref(self._h)), "lmx_create")
        check(lib.lmx_set_vocab_limit(self._h, int(config.vocab_size)), "lmx_set_vocab_limit")
        self.s_max = (c.max_position + 127) // 128 * 128
        self._set_rope_table()
        self.model = LlavaLlamaModel(self)
        self._lock = threading.Lock()
        self._tls = threading.local()       # per-request scratch (model_worker runs several generate threads on one model)
        self._batcher = None                # continuous-batching scheduler (enable_batching)
        self._prefill_gate = threading.Semaphore(1)
        self._batch_prefill_chunk = 0
        self._finalized = False
        self._img_cache = None              # reuse.ImageFeatureCache / reuse.PrefixCache (enable_reuse): off by default
        self._prefix = None

    @classmethod
    def from_pretrained(cls, model_path, *args, **kwargs):
        """llava/model/builder.py:100,106 call `LlavaLlamaForCausalLM.from_pretrained(path, low_cpu_mem_usage=True, config=..., **kwargs)`."""
        from .builder import from_pretrained
        return from_pretrained(model_path, *args, **kwargs)

    # ---- lifetime -----------------------------------------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.lmx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _set_rope_table(self):
        """cos/sin computed on the host exactly as LlamaRotaryEmbedding does (HF5:models/llama/modeling_llama.py:73-127)."""
        D = self.config.hidden_size // self.config.num_attention_heads
        theta = _rope_theta(self.config)
        inv_freq = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).to(torch.float32) / D))
        freqs = torch.arange(self.s_max, dtype=torch.float32)[:, None] * inv_freq[None, :]
        table = torch.cat([freqs.cos(), freqs.sin()], dim=-1).contiguous()
        check(lib.lmx_set_rope_table(self._h, ctypes.c_void_p(table.data_ptr()), self.s_max), "lmx_set_rope_table")

    # ---- weights --------------------------------------------------------------------------------------------------
    @staticmethod
    def canonical_name(key: str) -> Optional[str]:
        """Map a checkpoint key (LLaVA / HF-CLIP 4.31 / HF-CLIP 5.x layouts) to the engine's canonical name; None = skip."""
        if key.endswith("rotary_emb.inv_freq") or key.endswith("position_ids"):
            return None
        k = key
        for pre in ("model.vision_tower.vision_tower.", "vision_tower.vision_tower.", "vision_tower."):
            if k.startswith(pre):
                k = "vision::" + k[len(pre):]
                break
        if k.startswith("vision_model."):
            k = "vision::" + k
        if k.startswith("vision::"):
            k = k[len("vision::"):]
            if k.startswith("vision_model."):
                k = k[len("vision_model."):]
            if k.startswith("post_layernorm"):
                return None
            return "vision." + k
        if k.startswith("vision."):
            return None if "post_layernorm" in k else k
        if k.startswith("model.mm_projector."):
            return k[len("model."):]
        if k.startswith("embeddings.") or k.startswith("encoder.layers.") or k.startswith("pre_layrnorm."):
            return "vision." + k        # bare CLIPVisionModel keys (transformers 5.x)
        if k.startswith("post_layernorm"):
            return None
        return k

    def load_tensor(self, name: str, tensor: torch.Tensor) -> None:
        t = tensor.detach().to(device=self.device, dtype=self.dtype).contiguous()
        if name in ("model.embed_tokens.weight", "lm_head.weight") and t.shape[0] != self._vocab_cap:
            if t.shape[0] > self._vocab_cap:
                raise ValueError(f"{name} has {t.shape[0]} rows, the engine vocabulary holds {self._vocab_cap}")
            pad = torch.zeros((self._vocab_cap - t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], dim=0)           # padding ids: zero rows, masked out of every pick (lmx_set_vocab_limit)
        self._finalized = False
        shape = (ctypes.c_int64 * t.dim())(*t.shape)
        check(lib.lmx_load_weight(self._h, name.encode(), ptr(t), torch_dtype_code(self.dtype), t.dim(), shape, stream_handle()),
              f"lmx_load_weight({name})")
        torch.cuda.current_stream().synchronize()      # `t` may be a temporary; the copy must land before it is freed

    def load_state_dict(self, state: Mapping[str, torch.Tensor], strict: bool = True):
        for k, v in state.items():
            name = self.canonical_name(k)
            if name is None:
                continue
            self.load_tensor(name, v)
        if strict:
            self.finalize_weights()
        return self

    def finalize_weights(self):
        check(lib.lmx_finalize_weights(self._h), "lmx_finalize_weights")
        self._finalized = True

    def _ensure_final(self):
        """The reference has no explicit 'weights complete' step (from_pretrained + vision_tower.load_model()); the engine's
        completeness check therefore runs on first use."""
        if not self._finalized:
            self.finalize_weights()

    def init_tensor_parallel(self, force_comm: bool = False, rccl: bool = True, p2p: Optional[bool] = None):
        """Create the RCCL communicator: rank 0 makes the unique id, torch.distributed (any backend) broadcasts it.
        force_comm=True builds a 1-rank communicator for an unsharded model, so every decoder all-reduce site really
        calls ncclAllReduce on the launch stream (single-GPU test of the RCCL call path).
        p2p (default: on unless LLAVA_MI355X_P2P=0): also connect the one-shot peer-to-peer all-reduce used for decode-sized
        messages; it is self-tested on every rank and switched off everywhere unless all ranks pass.
        rccl=False (tests: two ranks sharing one GPU, which RCCL refuses) skips the communicator."""
        if self.tp_world == 1 and not force_comm:
            return
        import torch.distributed as dist
        if rccl:
            buf = (ctypes.c_uint8 * 128)()
            if self.tp_rank == 0:
                check(lib.lmx_tp_unique_id(buf), "lmx_tp_unique_id")
            obj = [bytes(buf)]
            if self.tp_world > 1:
                dist.broadcast_object_list(obj, src=0)
            raw = (ctypes.c_uint8 * 128).from_buffer_copy(obj[0])
            with torch.cuda.device(self.device):
                check(lib.lmx_tp_init(self._h, raw), "lmx_tp_init")
        if p2p is None:
            import os
            p2p = os.environ.get("LLAVA_MI355X_P2P", "1") != "0"
        self.p2p_active = False
        if p2p and self.tp_world > 1:
            ok, why = True, ""
            try:
                hd = (ctypes.c_uint8 * 64)()
                with torch.cuda.device(self.device):
                    check(lib.lmx_tp_p2p_local_handle(self._h, hd), "lmx_tp_p2p_local_handle")
                mine = bytes(hd)
            except Exception as e:  # noqa: BLE001
                ok, why, mine = False, repr(e), b"\0" * 64
            # every rank takes part in the SAME sequence of host collectives whatever happens locally (a rank that skipped one
            # after a local failure would leave the others waiting in it)
            gathered = [None] * self.tp_world
            dist.all_gather_object(gathered, (ok, mine))
            if all(g[0] for g in gathered):
                try:
                    blob = b"".join(g[1] for g in gathered)
                    with torch.cuda.device(self.device):
                        check(lib.lmx_tp_p2p_connect(self._h, (ctypes.c_uint8 * len(blob)).from_buffer_copy(blob)), "lmx_tp_p2p_connect")
                except Exception as e:  # noqa: BLE001
                    ok, why = False, repr(e)
            else:
                ok = False
            connected = [None] * self.tp_world
            dist.all_gather_object(connected, ok)
            if all(connected):
                try:
                    ok, why = self._p2p_selftest()
                except Exception as e:  # noqa: BLE001
                    ok, why = False, repr(e)
            else:
                ok = False
            verdicts = [None] * self.tp_world
            dist.all_gather_object(verdicts, (ok, why))
            self.p2p_active = all(v[0] for v in verdicts)
            if not self.p2p_active:
                try:
                    lib.lmx_tp_p2p_enable(self._h, 0)
                except Exception:  # noqa: BLE001
                    pass
                if self.tp_rank == 0:
                    import warnings
                    warnings.warn(f"peer-to-peer all-reduce disabled, decode all-reduces use RCCL: {[v[1] for v in verdicts if not v[0]]}")
                if not rccl:
                    raise RuntimeError(f"p2p all-reduce self-test failed and no RCCL communicator: {verdicts}")

    def tp_comm_ranks(self) -> int:
        """Ranks in the engine's RCCL communicator (ncclCommCount); 0 when there is none."""
        return int(lib.lmx_tp_comm_ranks(self._h))

    def _p2p_selftest(self) -> Tuple[bool, str]:
        """All-reduce integer-valued rows (exact in every dtype) through the P2P kernel and compare with the closed-form sum."""
        H, W, r = self.config.hidden_size, self.tp_world, self.tp_rank
        idx = torch.arange(H, device=self.device) % 13
        for it in range(12):
            for rows in (1, 4, 32) + ((33, 577) if it < 3 else ()):                                       # > 32 rows: the two-shot (reduce-scatter + all-gather) kernel
                base = (idx[None, :] + torch.arange(rows, device=self.device)[:, None] + it) % 7          # values 0..6
                mine = (base * (r + 1)).to(self.dtype).contiguous()
                want = (base * (W * (W + 1) // 2)).to(self.dtype)
                check(lib.lmx_op_allreduce(self._h, ptr(mine), rows * H, stream_handle()), "lmx_op_allreduce")
                torch.cuda.current_stream(self.device).synchronize()
                if lib.lmx_tp_p2p_status(self._h, stream_handle()) != 0:
                    return False, f"rank {r}: wait for a peer timed out (iteration {it}, rows {rows})"
                if not torch.equal(mine, want):
                    return False, f"rank {r}: wrong sum (iteration {it}, rows {rows})"
        return True, ""

    def set_option(self, key: str, value: int) -> None:
        """Flip one of the engine's launch-form options ("fuse_rope", "vis_pack", "decode_splitq": bit-identical forms; include/llava_mi355x.h)."""
        check(lib.lmx_model_set_option(self._h, key.encode(), int(value)), "lmx_model_set_option")

    # ---- in-situ kernel timing (HIP events on the launch stream) -----------------------------------------------------
    def profile(self, enable: bool) -> None:
        check(lib.lmx_profile_enable(self._h, int(enable)), "lmx_profile_enable")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        """{launch-group name: (total ms, launches)} accumulated since profile(True)."""
        names = ctypes.create_string_buffer(16384)
        ms = (ctypes.c_double * 256)(); cnt = (ctypes.c_int64 * 256)(); n = ctypes.c_int32(0)
        check(lib.lmx_profile_read(self._h, names, 16384, ms, cnt, 256, ctypes.byref(n)), "lmx_profile_read")
        keys = names.value.decode().split("\n")[: n.value]
        return {k: (ms[i], cnt[i]) for i, k in enumerate(keys)}

    # ---- reference API surface --------------------------------------------------------------------------------------
    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def half(self):
        return self

    def cuda(self, *a, **k):
        return self

    def requires_grad_(self, flag=False):
        return self

    def resize_token_embeddings(self, n: int):
        """builder.py:138 calls this after adding <im_patch> / <im_start> / <im_end>.  The engine tables were allocated with headroom
        (`vocab_headroom`), so growing or shrinking the REAL vocabulary inside that capacity is a limit change: new ids get zero
        embedding / lm_head rows (the reference leaves them untrained too), ids >= n are never picked and are sliced off the logits."""
        n = int(n)
        if n < 1 or n > self._vocab_cap:
            raise ValueError(f"resize_token_embeddings({n}): the engine vocabulary holds {self._vocab_cap} rows; construct the model "
                             f"with vocab_headroom >= {n - int(self.config.vocab_size)}")
        check(lib.lmx_set_vocab_limit(self._h, n), "lmx_set_vocab_limit")
        self.config.vocab_size = n
        return self.get_model().embed_tokens

    @property
    def tokens_per_image(self) -> int:
        return lib.lmx_tokens_per_image(self._h)

    def enable_reuse(self, images: int = 64, prefixes: int = 32, min_rows: int = 32) -> None:
        """Switch on the two caches of reuse.py: image features by pixel content (`images` entries of tokens_per_image x hidden each) and KV prefixes of
        finished requests (`prefixes` sequences: ~0.5 MB per position at 7B).  Off by default; bench.py never enables them.  Prefix reuse is a single-process
        feature (tensor-parallel followers replay the leader's calls; their caches would have to make the same decisions — not wired)."""
        from .reuse import ImageFeatureCache, PrefixCache
        self.disable_reuse()
        if images > 0:
            self._img_cache = ImageFeatureCache(images)
        if prefixes > 0 and self.tp_world == 1:
            self._prefix = PrefixCache(prefixes, min_rows)

    def disable_reuse(self) -> None:
        if self._prefix is not None:
            self._prefix.clear()
        self._img_cache, self._prefix = None, None

    def reuse_stats(self) -> dict:
        ic, pc = self._img_cache, self._prefix
        return {"image_hits": ic.hits if ic else 0, "image_misses": ic.misses if ic else 0, "prefix_hits": pc.hits if pc else 0,
                "prefix_misses": pc.misses if pc else 0, "prefix_rows_reused": pc.rows_reused if pc else 0, "prefix_entries": len(pc) if pc else 0}

    def _hash_images(self, x: torch.Tensor) -> List[Tuple[int, int]]:
        """128-bit content hash per image of a contiguous [N, 3, S, S] device tensor (lmx_op_hash128); one 16 N-byte read-back."""
        n = x.shape[0]
        out = torch.empty((n, 2), dtype=torch.int64, device=self.device)
        check(lib.lmx_op_hash128(ptr(x), x[0].numel() * x.element_size(), n, ptr(out), stream_handle()), "lmx_op_hash128")
        h = out.cpu().numpy().view(np.uint64)
        # dtype and geometry are part of the key: the same bytes under another interpretation are another image
        salt = hash((str(x.dtype), tuple(x.shape[1:]))) & 0xFFFFFFFF
        return [(int(h[i, 0]) ^ salt, int(h[i, 1])) for i in range(n)]

    def _check_pixels(self, images: torch.Tensor) -> torch.Tensor:
        if self.vision_config is None:
            raise ValueError("model has no vision tower")
        x = images.to(device=self.device, dtype=self.dtype).contiguous()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.vision_config.image_size or x.shape[3] != self.vision_config.image_size:
            raise ValueError(f"images must be [N,3,{self.vision_config.image_size},{self.vision_config.image_size}], got {tuple(x.shape)}")
        return x

    def _run_tower(self, x: torch.Tensor, out: torch.Tensor, sharded: bool) -> None:
        """Tower + projector over x [n,3,S,S] into out [n,P,H].  sharded (tensor parallel, n >= 2; a COLLECTIVE — every rank calls it with the same
        pixels at the same point of its call sequence): the tower is data parallel over the images ("split images across ranks for B > 1,
        then all-gather features") — rank r runs images r, r + W, ..., the other rows of `out` stay zero and the decoder's all-reduce sums the ranks'
        buffers, which IS the all-gather (every element has one non-zero contributor: exact in every dtype, like the vocabulary-parallel logits row)."""
        n = x.shape[0]
        if not sharded:
            check(lib.lmx_encode_images(self._h, ptr(x), n, ptr(out), stream_handle()), "lmx_encode_images")
            return
        mine = torch.arange(self.tp_rank, n, self.tp_world, device=self.device)
        out.zero_()
        if mine.numel():
            xm = x.index_select(0, mine).contiguous()
            om = torch.empty((int(mine.numel()),) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
            check(lib.lmx_encode_images(self._h, ptr(xm), int(mine.numel()), ptr(om), stream_handle()), "lmx_encode_images")
            out.index_copy_(0, mine, om)
        # pieces of <= 4096 rows of H elements: what the two-shot peer-to-peer all-reduce takes in one launch (engine.cpp: p2p_big_max_count)
        per = max(1, 4096 // max(1, out.shape[1] * out.shape[2] // self.config.hidden_size))
        for i0 in range(0, n, per):
            piece = out[i0:i0 + per]
            check(lib.lmx_op_allreduce(self._h, ptr(piece), piece.numel(), stream_handle()), "lmx_op_allreduce")

    def tower_is_sharded(self, n_images: int) -> bool:
        """Whether encode_images_sharded() would split n_images over the tensor-parallel ranks (LLAVA_MI355X_TP_TOWER=0 keeps the tower replicated)."""
        import os
        return self.tp_world > 1 and n_images >= 2 and os.environ.get("LLAVA_MI355X_TP_TOWER", "1") != "0"

    def encode_images_sharded(self, images: torch.Tensor) -> torch.Tensor:
        """encode_images for the images of a whole BATCH of requests under tensor parallelism: the ranks split the images and all-gather the features
        (_run_tower).  COLLECTIVE: every rank must call it with the same pixels, in step with its other engine calls — generate_batch and
        tp_serving.prefill_symmetric do; a request thread's own encode_images never is (the tower stays replicated there)."""
        return self.encode_images(images, _collective=True)

    def encode_images(self, images: torch.Tensor, _collective: bool = False) -> torch.Tensor:
        """llava_arch.py:94-97 — images [N,3,S,S] -> [N, tokens_per_image, hidden].  With enable_reuse(): images whose pixels were encoded before return
        the stored rows (reuse.ImageFeatureCache), only the others run the tower."""
        ov = getattr(self._tls, "feats_override", None)
        if ov is not None:
            # features of this request's images were computed with the batch's (generate_batch / prefill_symmetric: encode_images_sharded)
            self._tls.feats_override = None
            feats, hashes = ov
            if feats.shape[0] != images.shape[0]:
                raise ValueError(f"pre-encoded features cover {feats.shape[0]} images, the request has {images.shape[0]}")
            self._tls.image_hashes = hashes
            return feats
        x = self._check_pixels(images)
        self._ensure_final()
        n = x.shape[0]
        P = self.tokens_per_image
        out = torch.empty((n, P, self.config.hidden_size), dtype=self.dtype, device=self.device)
        cache = self._img_cache
        self._tls.image_hashes = None
        if cache is None or n == 0:
            self._run_tower(x, out, _collective and self.tower_is_sharded(n))
            return out
        hashes = self._hash_images(x)
        self._tls.image_hashes = hashes
        stored = [cache.get(h) for h in hashes]
        miss = [i for i in range(n) if stored[i] is None]
        if miss:
            # the caches of all ranks hold the same entries (same calls in the same order), so the miss list — and with it the split — is the same everywhere
            sharded = _collective and self.tower_is_sharded(len(miss))
            if len(miss) == n:
                self._run_tower(x, out, sharded)
                fresh = out
            else:
                xm = x.index_select(0, torch.tensor(miss, device=self.device)).contiguous()
                fresh = torch.empty((len(miss), P, self.config.hidden_size), dtype=self.dtype, device=self.device)
                self._run_tower(xm, fresh, sharded)
            for j, i in enumerate(miss):
                if fresh is not out:
                    out[i].copy_(fresh[j])
                cache.put(hashes[i], fresh[j].clone())
        for i in range(n):
            if stored[i] is not None:
                out[i].copy_(stored[i])
        return out

    def _preencode_local(self, images_per_request: Sequence):
        """Rank-local half of the sharded tower pass over the images of several requests (tensor parallel): geometry checks, the concatenation, the output
        buffer and THIS rank's share of the images through the tower — everything that can fail on one rank alone (a device copy, an allocation), and
        nothing collective.  The split is a function of the image count only (rank r takes images r, r + W, ...; the image-feature cache is NOT consulted
        here: a cache that diverged between ranks must not change who encodes what).  Returns None when there is nothing to shard, else the state for
        _preencode_collective.  images_per_request: per request a [k,3,S,S] tensor, None, or anything else (left to the request's own, replicated, encode)."""
        idx = [i for i, im in enumerate(images_per_request) if isinstance(im, torch.Tensor) and im.dim() == 4 and im.shape[0] > 0]
        if not idx or not self.tower_is_sharded(sum(int(images_per_request[i].shape[0]) for i in idx)):
            return None
        self._ensure_final()
        cat = torch.cat([self._check_pixels(images_per_request[i]) for i in idx], dim=0)
        n = int(cat.shape[0])
        out = torch.zeros((n, self.tokens_per_image, self.config.hidden_size), dtype=self.dtype, device=self.device)
        mine = torch.arange(self.tp_rank, n, self.tp_world, device=self.device)
        if mine.numel():
            xm = cat.index_select(0, mine).contiguous()
            om = torch.empty((int(mine.numel()),) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
            check(lib.lmx_encode_images(self._h, ptr(xm), int(mine.numel()), ptr(om), stream_handle()), "lmx_encode_images")
            out.index_copy_(0, mine, om)
        return {"idx": idx, "counts": [int(images_per_request[i].shape[0]) for i in idx], "n_req": len(images_per_request), "cat": cat, "out": out}

    def _preencode_collective(self, st) -> List:
        """COLLECTIVE half: the ranks' buffers are summed (every element has one non-zero contributor: the sum IS the all-gather, exact in every dtype).
        Only launches all-reduces on buffers that already exist.  Returns per request None or the (features [k,P,H], image hashes) pair that
        `_tls.feats_override` hands to the request's encode_images; with enable_reuse() the features also enter the image-feature cache."""
        out = st["out"]
        n = int(out.shape[0])
        # pieces of <= 4096 rows of H elements: what the two-shot peer-to-peer all-reduce takes in one launch (engine.cpp: p2p_big_max_count)
        per = max(1, 4096 // max(1, out.shape[1] * out.shape[2] // self.config.hidden_size))
        for i0 in range(0, n, per):
            piece = out[i0:i0 + per]
            check(lib.lmx_op_allreduce(self._h, ptr(piece), piece.numel(), stream_handle()), "lmx_op_allreduce")
        hashes = None
        if self._img_cache is not None:
            hashes = self._hash_images(st["cat"])
            for i, h in enumerate(hashes):
                if self._img_cache.get(h) is None:
                    self._img_cache.put(h, out[i].clone())
        res: List = [None] * st["n_req"]
        o = 0
        for i, k in zip(st["idx"], st["counts"]):
            res[i] = (out[o:o + k], None if hashes is None else hashes[o:o + k])
            o += k
        return res

    def _preencode_requests(self, images_per_request: Sequence) -> List:
        """The images of several requests through ONE sharded tower pass (tensor parallel; a COLLECTIVE): rank-local half + collective half back to back,
        for callers whose ranks run in lock step by construction (generate_batch).  The serving path brackets the halves with agreements
        (tp_serving.prefill_symmetric)."""
        st = self._preencode_local(images_per_request)
        return [None] * len(images_per_request) if st is None else self._preencode_collective(st)

    def clear_image_cache(self) -> None:
        """Drop every stored image feature (after a failure that may have left the tensor-parallel ranks' caches different)."""
        if self._img_cache is not None:
            self._img_cache.clear()

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images):
        """llava_arch.py:99-240, same 6-tuple.  Integer half on the host through lmx_splice_plan (bit-exact), embedding
        gather/splice on the device."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            if past_key_values is not None and vision_tower is not None and images is not None and input_ids.shape[1] == 1:
                target_shape = past_key_values[-1][-1].shape[-2] + 1                  # llava_arch.py:105
                attention_mask = torch.cat((attention_mask, torch.ones((attention_mask.shape[0], target_shape - attention_mask.shape[1]),
                                                                       dtype=attention_mask.dtype, device=attention_mask.device)), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels

        if type(images) is list or images.ndim == 5:                                   # llava_arch.py:114-119
            concat = torch.cat([im for im in images], dim=0)
            feats = self.encode_images(concat)
            P = feats.shape[1]
            slot_rows = np.asarray([im.shape[0] * P for im in images], dtype=np.int32)
            feat_mat = feats.reshape(-1, feats.shape[-1])
        else:
            feats = self.encode_images(images)
            P = feats.shape[1]
            slot_rows = None
            feat_mat = feats.reshape(-1, feats.shape[-1])
        n_slots = len(images) if slot_rows is not None else feats.shape[0]

        if getattr(self.config, "tune_mm_mlp_adapter", False) and getattr(self.config, "mm_use_im_start_end", False):
            raise NotImplementedError                                                   # llava_arch.py:124-125

        ids_h = np.ascontiguousarray(input_ids.detach().cpu().numpy().astype(np.int64))        # same D2H the reference does (:161)
        B, L = ids_h.shape
        mask_h = None if attention_mask is None else np.ascontiguousarray(attention_mask.detach().cpu().numpy().astype(np.uint8))
        lab_h = None if labels is None else np.ascontiguousarray(labels.detach().cpu().numpy().astype(np.int64))
        max_len = getattr(self.config, "tokenizer_model_max_length", None) or 0
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        vp = lambda a: ctypes.c_void_p(0) if a is None else ctypes.c_void_p(a.ctypes.data)
        T = ctypes.c_int32(0)
        args = (vp(ids_h), vp(mask_h), vp(lab_h), B, L, P, vp(slot_rows), n_slots, int(max_len), int(left))
        rc = lib.lmx_splice_plan(*args, ctypes.byref(T), None, None, None, None)
        if rc:
            raise IndexError(_C.last_error())          # the reference raises IndexError on image_features[cur_image_idx]
        Tn = T.value
        src = np.empty((B, Tn), np.int32); om = np.empty((B, Tn), np.uint8)
        op = np.empty((B, Tn), np.int64); ol = np.empty((B, Tn), np.int64)
        check(lib.lmx_splice_plan(*args, ctypes.byref(T), vp(src), vp(om), vp(op), vp(ol)), "lmx_splice_plan")
        if (src >= self.config.vocab_size).any():
            raise IndexError("token id out of range for embed_tokens")
        src_d = torch.from_numpy(src).to(self.device)
        embeds = torch.empty((B, Tn, self.config.hidden_size), dtype=self.dtype, device=self.device)
        check(lib.lmx_gather_embeds(self._h, ptr(src_d), B * Tn, ptr(feat_mat.to(self.dtype).contiguous()), ptr(embeds), stream_handle()), "lmx_gather_embeds")
        dev = input_ids.device
        new_labels = None if labels is None else torch.from_numpy(ol).to(device=labels.device, dtype=labels.dtype)
        new_mask = None if attention_mask is None else torch.from_numpy(om).to(device=attention_mask.device, dtype=attention_mask.dtype)
        new_pos = None if position_ids is None else torch.from_numpy(op).to(device=position_ids.device, dtype=position_ids.dtype)
        # keep the plan's mask for the decoder even when the caller passed attention_mask=None
        self._tls.plan_mask = torch.from_numpy(om.astype(bool))
        self._tls.plan_src, self._tls.plan_P = src, P           # row identities for the prefix cache (reuse.row_keys)
        return None, new_pos, new_mask, past_key_values, embeds, new_labels

    # ---- decoder ---------------------------------------------------------------------------------------------------
    def _prefill_rows(self, cache: LmxKVCache, embeds: torch.Tensor, valid: Optional[torch.Tensor], want_all: bool, greedy: bool, chunk: int = 0):
        """Run the decoder over [B,T,H] embeddings; `valid` [B,T] bool marks real (non-pad) positions."""
        B, T, H = embeds.shape
        V = self._vocab_cap                 # row pitch of the engine's logits; callers slice to config.vocab_size
        logits = torch.zeros((B, T if want_all else 1, V), dtype=self.dtype, device=self.device)
        for b in range(B):
            if valid is not None:
                idx = torch.nonzero(valid[b].to(self.device), as_tuple=False).flatten()
                n = int(idx.numel())
                if n == 0:
                    continue
                e = embeds[b] if n == T else embeds[b].index_select(0, idx).contiguous()
            else:
                idx, n, e = None, T, embeds[b]
            e = e.contiguous()
            if want_all:
                lg = logits[b] if n == T else torch.empty((n, V), dtype=self.dtype, device=self.device)
                check(lib.lmx_prefill(self._h, cache.seqs[b], ptr(e), n, chunk, ptr(lg), 1, int(greedy), stream_handle()), "lmx_prefill")
                if n != T:
                    logits[b].index_copy_(0, idx, lg)
            else:
                check(lib.lmx_prefill(self._h, cache.seqs[b], ptr(e), n, chunk, ptr(logits[b]), 0, int(greedy), stream_handle()), "lmx_prefill")
        return logits

    def _prefill_rows_outputs(self, cache: LmxKVCache, embeds: torch.Tensor, valid: Optional[torch.Tensor], want_hidden: bool, want_attn: bool):
        """_prefill_rows(want_all=True) that also returns `output_hidden_states`' tuple — L + 1 tensors [B,T,H]: entry l < L = the rows entering decoder layer l,
        entry L = the final norm's output (HF5:models/llama/modeling_llama.py:367-418) — and / or `output_attentions`' tuple — L tensors [B, heads, T, past + T]:
        each layer's post-softmax attention weights as the eager path returns them (:191-214), recomputed from the rotated q rows and the K cache
        (lmx_prefill_outputs).  Pad rows / pad key columns stay zero."""
        B, T, H = embeds.shape
        V, L, nh = self._vocab_cap, self.config.num_hidden_layers, self.config.num_attention_heads
        past = cache.lengths()
        logits = torch.zeros((B, T, V), dtype=self.dtype, device=self.device)
        hidden = torch.zeros((L + 1, B, T, H), dtype=self.dtype, device=self.device) if want_hidden else None
        attn = None
        if want_attn:
            if self.tp_world > 1:
                raise NotImplementedError("output_attentions under tensor parallelism: every rank holds a slice of the heads")
            if any(p > 0 for p in past) and valid is not None and not bool(valid.all()):
                raise NotImplementedError("output_attentions with padding on top of a KV cache is not supported (un-padded continuation, or padded batches from an empty cache)")
            attn = torch.zeros((L, B, nh, T, max(past) + T), dtype=self.dtype, device=self.device)
        for b in range(B):
            idx = None
            if valid is not None:
                idx = torch.nonzero(valid[b].to(self.device), as_tuple=False).flatten()
                if int(idx.numel()) == T:
                    idx = None
                elif int(idx.numel()) == 0:
                    continue
            e = (embeds[b] if idx is None else embeds[b].index_select(0, idx)).contiguous()
            n = e.shape[0]
            lg = torch.empty((n, V), dtype=self.dtype, device=self.device)
            hs = torch.empty((L + 1, n, H), dtype=self.dtype, device=self.device) if want_hidden else None
            ap = torch.empty((L, nh, n, past[b] + n), dtype=self.dtype, device=self.device) if want_attn else None
            check(lib.lmx_prefill_outputs(self._h, cache.seqs[b], ptr(e), n, ptr(lg), 1, ptr(hs), ptr(ap), stream_handle()), "lmx_prefill_outputs")
            if idx is None:
                logits[b] = lg
                if want_hidden:
                    hidden[:, b] = hs
                if want_attn:
                    attn[:, b, :, :, : past[b] + n] = ap
            else:
                logits[b].index_copy_(0, idx, lg)
                if want_hidden:
                    hidden[:, b].index_copy_(1, idx, hs)
                if want_attn:
                    attn[:, b][:, :, idx[:, None], idx[None, :]] = ap           # past == 0 here: compacted rows / keys back to their padded places
        return (logits, None if hidden is None else tuple(hidden[l] for l in range(L + 1)), None if attn is None else tuple(attn[l] for l in range(L)))

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None, **kwargs):
        """llava_llama.py:56-99.  Returns CausalLMOutputWithPast(loss, logits [B,T,V] fp32 (as transformers 4.31 does),
        past_key_values=LmxKVCache).  Pad positions (attention_mask == 0) get zero logits.  output_hidden_states=True (llava_llama.py:63-64)
        adds `hidden_states`, LlamaModel's tuple of L + 1 tensors [B,T,H] (pad rows zero); output_attentions=True (llava_llama.py:62-63) adds `attentions`, the
        tuple of L tensors [B, heads, T, past + T] the eager attention returns (pad rows / columns zero) — recomputed beside the fused kernels for this call only."""
        hidden_states, attentions = None, None
        want_out = bool(output_hidden_states) or bool(output_attentions)
        self._ensure_final()
        user_pos = position_ids
        plan_mask = None
        if inputs_embeds is None:
            self._tls.plan_mask = None
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, past_key_values, labels, images)
            plan_mask = self._tls.plan_mask
        if inputs_embeds is None:
            inputs_embeds = self.get_model().embed_tokens(input_ids)
        inputs_embeds = inputs_embeds.to(device=self.device, dtype=self.dtype)
        B, T, _ = inputs_embeds.shape
        if past_key_values is not None and not isinstance(past_key_values, LmxKVCache):
            raise TypeError("past_key_values must come from this model (LmxKVCache)")
        decode_step = past_key_values is not None and T == 1 and input_ids is not None
        if decode_step:
            cache = past_key_values
            V = self._vocab_cap
            logits = torch.empty((B, 1, V), dtype=self.dtype, device=self.device)
            toks = input_ids.reshape(-1).tolist()
            if position_ids is not None:
                want = torch.tensor(cache.lengths(), dtype=torch.long)
                if not torch.equal(position_ids.reshape(-1).cpu().long(), want):
                    raise ValueError("position_ids must continue each sequence's cache (the fused RoPE derives positions from the KV-cache length)")
            if want_out:
                # one position through the prefill path, which can hand out the layer inputs / attention rows (the decode kernels keep them in their workspace only)
                logits, hidden_states, attentions = self._prefill_rows_outputs(cache, inputs_embeds, None, bool(output_hidden_states), bool(output_attentions))
            else:
                for b in range(B):
                    check(lib.lmx_decode(self._h, cache.seqs[b], int(toks[b]), 1, ptr(logits[b]), 0, stream_handle()), "lmx_decode")
        else:
            cache = past_key_values if past_key_values is not None else LmxKVCache(self, B)
            valid = None
            if attention_mask is not None:
                valid = attention_mask[:, -T:].bool()
            elif plan_mask is not None:
                valid = plan_mask
            # position_ids are implied by the mask (consecutive over each row's unmasked tokens, exactly what the splice
            # builds at llava_arch.py:206-223 and what HF derives when none are given); anything else is refused, not ignored.
            if position_ids is not None:
                base = torch.tensor(cache.lengths(), dtype=torch.long)[:, None]
                ok = torch.ones((B, T), dtype=torch.bool) if valid is None else valid.cpu().bool()
                want = (torch.cumsum(ok.long(), dim=1) - 1).clamp_min(0) + base
                got = position_ids[:, -T:].cpu().long().expand(B, T)
                if not torch.equal(got[ok], want[ok]):
                    raise ValueError("non-consecutive position_ids are not supported: the fused RoPE numbers each row's unmasked tokens 0, 1, 2, ...")
            if want_out:
                logits, hidden_states, attentions = self._prefill_rows_outputs(cache, inputs_embeds, valid, bool(output_hidden_states), bool(output_attentions))
            else:
                logits = self._prefill_rows(cache, inputs_embeds, valid, want_all=True, greedy=False)
        logits = logits[..., : self.config.vocab_size].float()        # padded ids (engine row pitch) are not part of the vocabulary
        loss = None
        if labels is not None:
            # LlamaForCausalLM's shifted cross-entropy (logits[..., :-1, :] vs labels[..., 1:], IGNORE_INDEX skipped, mean) on the
            # device: csrc/train.hip ce_fwd_kernel, fp32 math on the fp32 logits exactly as transformers 4.31 upcasts them
            from . import ops
            lab = labels.to(device=logits.device, dtype=torch.long).contiguous()
            loss, _, _ = ops.ce_loss(logits.contiguous(), lab, ignore_index=IGNORE_INDEX)
        if use_cache is False and past_key_values is None:
            cache.close()
            cache = None
        if return_dict is False:
            return tuple(x for x in (loss, logits, cache, hidden_states, attentions) if x is not None)
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache, hidden_states=hidden_states, attentions=attentions)

    __call__ = forward

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        """llava_llama.py:101-108: HF's default slicing + re-attached `images`."""
        images = kwargs.pop("images", None)
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        out = {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
               "attention_mask": kwargs.get("attention_mask")}
        if inputs_embeds is not None and past_key_values is None:
            out = {"inputs_embeds": inputs_embeds, **{k: v for k, v in out.items() if k != "input_ids"}}
        if images is not None:
            out["images"] = images
        return out

    # ---- continuous batching --------------------------------------------------------------------------------
    def enable_batching(self, capacity: int = 32, prefill_chunk: int = 0, prewarm: bool = True, channel=None, packed_prefill: bool = True,
                        max_prefill_batch: int = 8, max_prefill_rows: int = 2304, prefill_thread: int = 2) -> None:
        """From now on concurrent generate() calls (model_worker.py:174-185 runs one thread per request) decode together:
        one scheduler thread steps every live request through lmx_decode_batch.  packed_prefill (default): the scheduler also prefills — the requests
        waiting at that moment (up to max_prefill_batch) go through lmx_prefill_batch together, between two decode steps; packed_prefill=False keeps
        each prefill on its request's own thread and stream (one at a time), where the running decode batch interleaves with it at kernel
        granularity.  `max_prefill_rows` bounds the rows of one packed prefill (default 2304: two single-image 512-token requests), i.e. how long the live requests
        wait between two of their decode steps (~18 ms per 1k rows at 7B) and the packed workspace; 0 = up to max_prefill_batch requests whatever
        their size (best burst throughput, unbounded stall).  `prefill_chunk` > 0 bounds the rows per prefill piece (workspace; costs GEMM efficiency).
        prefill_thread (single process only) = 2 (default): the packed prefills run on their own thread and high-priority stream beside the decode steps instead
        of between two of them — a new request does not wait behind the step in flight, the live requests do not wait for its prefill — and each request's
        rank-local half (image encode / feature-cache hit, splice, prefix-cache take) runs on the request's own thread before it queues; 1 = that half on the
        prefill thread too; 0 = round 4's order (one loop: prefill, then step)."""
        from .batching import DecodeBatcher
        if self.tp_world > 1:
            # tensor parallel: only the leader schedules, and it needs the command channel to its followers (tp_serving.py)
            if self.tp_rank != 0:
                raise RuntimeError("tensor-parallel rank > 0 does not schedule: run tp_serving.serve_follower(model, channel)")
            if channel is None:
                raise RuntimeError("enable_batching under tensor parallelism needs channel=tp_serving.CommandChannel(group)")
        if self._batcher is None:
            self._ensure_final()
            self._batcher = DecodeBatcher(self, capacity, channel=channel if self.tp_world > 1 else None, scheduler_prefill=bool(packed_prefill),
                                          max_prefill_batch=int(max_prefill_batch), max_prefill_rows=int(max_prefill_rows), prefill_thread=int(prefill_thread))
            self._batch_prefill_chunk = int(prefill_chunk)
            if prewarm:
                # allocate (and zero) the KV caches of `capacity` sequences now; closing them parks them in the engine's sequence
                # pool, so the first burst of requests does not pay hipMalloc + a 1 GB memset each in front of its prefill
                warm = [LmxKVCache(self, 1) for _ in range(int(capacity))]
                for c in warm:
                    c.close()

    def disable_batching(self) -> None:
        if self._batcher is not None:
            self._batcher.close()
            self._batcher = None

    @torch.no_grad()
    def generate_batch(self, prompts, images=None, max_new_tokens: int = 20, eos_token_id=None, run_ahead: int = 16,
                       prefill_chunk: int = 0, capacity: Optional[int] = None, attention_masks=None, do_sample: bool = False,
                       temperature: float = 1.0, top_p: Optional[float] = None, top_k: Optional[int] = None, packed_prefill: bool = True):
        """Offline batch generation (greedy, or sampled on the device): the requests are prefilled TOGETHER (own image, own prompt length — no padding: their
        rows form one packed block walked in pieces of prefill_chunk x requests rows, every linear a single GEMM per piece; lmx_prefill_batch), then
        all of them decode together, `run_ahead` chained steps per host round trip; finished requests leave the batch.  packed_prefill=False prefills
        request by request (lmx_prefill, `prefill_chunk` rows at a time).
        prompts: list of LongTensor [L_i] / [1, L_i] (with -200 markers); images: list of per-request tensors or None.
        Returns a list of LongTensor [L_i + new_i] (input ids echoed, like generate())."""
        from .batching import DecodeBatch
        self._ensure_final()
        n_req = len(prompts)
        images = images if images is not None else [None] * n_req
        eos = eos_token_id if eos_token_id is not None else getattr(self.config, "eos_token_id", None)
        eos_set = set(eos if isinstance(eos, (list, tuple)) else ([eos] if eos is not None else []))
        caches, outs, budgets, packed = [], [], [], []
        batch = DecodeBatch(self, capacity or max(1, n_req))
        try:
            greedy = (not do_sample) or (temperature is not None and temperature <= 1e-5)
            masks = attention_masks if attention_masks is not None else [None] * n_req
            # tensor parallel: the images of ALL requests through one tower pass split over the ranks (every rank runs this same plan)
            pre = self._preencode_requests(images) if self.tp_world > 1 else [None] * n_req
            for ids, img, am, ov in zip(prompts, images, masks, pre):
                ids = ids if ids.dim() == 2 else ids[None]
                am = None if am is None else (am if am.dim() == 2 else am[None])
                self._tls.plan_mask = None
                self._tls.feats_override = ov
                try:
                    _, _, mask, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(ids, None, am, None, None, img)
                finally:
                    self._tls.feats_override = None
                if embeds is None:
                    embeds = self.get_model().embed_tokens(ids.to(self.device)); valid = None if am is None else am.bool()
                else:
                    valid = self._tls.plan_mask if mask is None else mask.bool()
                cache = LmxKVCache(self, 1)
                caches.append(cache)
                self._set_stop(cache.seqs[0], self._stop_spec(eos_set))     # a member that reached EOS stops advancing inside the batched steps
                if not greedy:
                    seed = self._draw_seed()
                    check(lib.lmx_seq_set_sampling(cache.seqs[0], float(temperature), float(top_p if top_p is not None else 1.0), int(top_k or 0), seed),
                          "lmx_seq_set_sampling")
                if packed_prefill:
                    # rows of this request for the packed prefill: its valid (non-pad) positions, in order
                    e = embeds[0]
                    if valid is not None and not bool(valid[0].all()):
                        e = e.index_select(0, torch.nonzero(valid[0].to(self.device), as_tuple=False).flatten())
                    packed.append(e.to(self.dtype).contiguous())
                else:
                    self._prefill_rows(cache, embeds, valid, want_all=False, greedy=True, chunk=prefill_chunk)
            if packed_prefill:
                n = len(packed)
                arr = (ctypes.c_void_p * n)(*[c.seqs[0].value if isinstance(c.seqs[0], ctypes.c_void_p) else c.seqs[0] for c in caches])
                eptr = (ctypes.c_void_p * n)(*[e.data_ptr() for e in packed])
                cnt = (ctypes.c_int32 * n)(*[int(e.shape[0]) for e in packed])
                check(lib.lmx_prefill_batch(self._h, arr, n, eptr, cnt, int(prefill_chunk) * n if prefill_chunk else 0, 1, stream_handle()), "lmx_prefill_batch")
            for cache in caches:
                budgets.append(min(max_new_tokens, self.s_max - lib.lmx_seq_length(cache.seqs[0])))
                host1 = (ctypes.c_int64 * 1)(); n1 = ctypes.c_int32(0)
                check(lib.lmx_seq_read_tokens(cache.seqs[0], host1, 1, ctypes.byref(n1), stream_handle()), "read_tokens")
                outs.append([int(host1[0])] if budgets[-1] > 0 else [])
            live = [i for i in range(n_req) if budgets[i] > 1 and not (outs[i] and outs[i][-1] in eos_set)]
            while live:
                for g0 in range(0, len(live), batch.capacity):
                    grp = live[g0:g0 + batch.capacity]
                    k = max(1, min([run_ahead] + [budgets[i] - len(outs[i]) for i in grp]))
                    ids_steps = batch.step([caches[i].seqs[0] for i in grp], None, k, True)
                    for j, i in enumerate(grp):
                        for st in range(k):
                            if len(outs[i]) >= budgets[i] or (outs[i] and outs[i][-1] in eos_set) or ids_steps[st][j] < 0:
                                break
                            outs[i].append(ids_steps[st][j])
                live = [i for i in live if len(outs[i]) < budgets[i] and outs[i][-1] not in eos_set]
            res = []
            for ids, o in zip(prompts, outs):
                flat = ids.reshape(-1).cpu()
                res.append(torch.cat([flat, torch.tensor(o, dtype=torch.long)]).to(self.device))
            return res
        finally:
            batch.close()
            for c in caches:
                c.close()

    # ---- generation ------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def generate(self, inputs=None, images=None, do_sample=False, temperature=1.0, top_p=None, top_k=None, num_beams=1,
                 max_new_tokens=None, max_length=None, streamer=None, stopping_criteria=None, use_cache=True, attention_mask=None,
                 eos_token_id=None, pad_token_id=None, input_ids=None, run_ahead: int = 16, prefill_chunk: int = 0, **kwargs):
        """Greedy / temperature+top-p generation with the device-resident decode loop (model_worker.py:174-185 contract:
        returns LongTensor [B, L + new] that echoes the input ids, image markers included)."""
        if inputs is None:
            inputs = input_ids
        if inputs is None:
            raise ValueError("generate() needs `inputs` (input_ids)")
        self._ensure_final()
        if not use_cache:
            raise NotImplementedError("generate() always uses the KV cache")
        ids = inputs if inputs.dim() == 2 else inputs[None]
        B, L = ids.shape
        if max_new_tokens is None:
            max_new_tokens = (max_length - L) if max_length is not None else 20
        eos = eos_token_id if eos_token_id is not None else getattr(self.config, "eos_token_id", None)
        eos_set = set(eos if isinstance(eos, (list, tuple)) else ([eos] if eos is not None else []))
        pad = pad_token_id if pad_token_id is not None else (getattr(self.config, "pad_token_id", None) or 0)
        greedy = (not do_sample) or (temperature is not None and temperature <= 1e-5)
        rows: List[List[int]] = []
        if num_beams != 1:
            # GenerationMixin.beam_search / beam_sample + BeamSearchScorer (llava_mi355x/beam.py); transformers refuses a streamer here as well
            if streamer is not None:
                raise ValueError("`streamer` cannot be used with beam search (as in transformers)")
            from .beam import beam_search
            beam_images = self._row_images(ids, images, attention_mask)      # row b's own images (llava_arch.py:150-159 slot arithmetic)
            for b in range(B):
                row_mask = None if attention_mask is None else attention_mask[b:b + 1]
                rows.append(beam_search(self, ids[b:b + 1], beam_images[b], row_mask, int(num_beams), int(max_new_tokens), eos_set, float(kwargs.get("length_penalty", 1.0)),
                                        kwargs.get("early_stopping", False), prefill_chunk, bool(kwargs.get("length_counts_prompt", True)),
                                        eos_first=(eos[0] if isinstance(eos, (list, tuple)) and eos else (eos if isinstance(eos, int) and eos >= 0 else None)),
                                        stopping_criteria=list(stopping_criteria) if stopping_criteria else None,
                                        # num_beams > 1 with do_sample=True: GenerationMixin.beam_sample (beam.py)
                                        sample=None if greedy else dict(temperature=float(temperature), top_p=top_p, top_k=top_k, seed=self._draw_seed())))
            width = L + max(len(r) for r in rows)
            out = torch.full((B, width), pad, dtype=torch.long)
            for b, r in enumerate(rows):
                out[b, :L] = ids[b].cpu()
                out[b, L:L + len(r)] = torch.tensor(r, dtype=torch.long)
            return out.to(ids.device)
        if streamer is not None:
            if B != 1:
                raise ValueError("streaming needs batch size 1")
            streamer.put(ids.cpu())
        row_images = self._row_images(ids, images, attention_mask)
        if B > 1 and streamer is None and not stopping_criteria and max_new_tokens > 0:
            # rows decode together (one pass over the weights per step for the whole batch)
            outs = self.generate_batch([ids[b] for b in range(B)], row_images, max_new_tokens=max_new_tokens, eos_token_id=list(eos_set) or -1,
                                       run_ahead=run_ahead, prefill_chunk=prefill_chunk,
                                       attention_masks=None if attention_mask is None else [attention_mask[b] for b in range(B)],
                                       do_sample=not greedy, temperature=temperature, top_p=top_p, top_k=top_k)
            rows = [o[L:].tolist() for o in outs]
        else:
            for b in range(B):
                row_mask = None if attention_mask is None else attention_mask[b:b + 1]
                rows.append(self._generate_one(ids[b:b + 1], row_images[b], row_mask, greedy, temperature, top_p, top_k, max_new_tokens,
                                               eos_set, streamer, stopping_criteria, run_ahead, prefill_chunk))
        if streamer is not None:
            streamer.end()
        if self.tp_world > 1 and getattr(self, "p2p_active", False):
            st = lib.lmx_tp_p2p_status(self._h, stream_handle())
            if st != 0:
                raise RuntimeError(f"tensor-parallel rank {self.tp_rank}: peer-to-peer all-reduce #{st} timed out waiting for a peer")
        width = L + max(len(r) for r in rows)
        out = torch.full((B, width), pad, dtype=torch.long)
        for b, r in enumerate(rows):
            out[b, :L] = ids[b].cpu()
            out[b, L:L + len(r)] = torch.tensor(r, dtype=torch.long)
        return out.to(ids.device)

    @staticmethod
    def _row_images(ids, images, attention_mask) -> List:
        """A batch shares one `images` argument: row b owns the next max(1, #markers) entries, the slot arithmetic of llava_arch.py:150-159 (a text-only
        row still consumes one slot)."""
        B = ids.shape[0]
        if images is None or B == 1:
            return [images] * B
        row_images, nxt = [], 0
        for b in range(B):
            valid_ids = ids[b] if attention_mask is None else ids[b][attention_mask[b].bool().to(ids.device)]
            n_img = max(1, int((valid_ids == IMAGE_TOKEN_INDEX).sum().item()))
            row_images.append(images[nxt:nxt + n_img]); nxt += n_img
        return row_images

    def _draw_seed(self) -> int:
        """Seed of a request's device sampler, from torch's CPU generator (torch.manual_seed makes a request reproducible).  Under
        tensor parallelism every rank must draw the SAME token from the same logits — ranks that diverge feed different ids into the
        shared all-reduces — so rank 0's seed is broadcast (one more call in the identical host-collective sequence of the ranks)."""
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if self.tp_world > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("sampled generation under tensor parallelism needs torch.distributed (rank 0's sampler seed is broadcast)")
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
        return seed

    @staticmethod
    def _stop_spec(eos_set, criteria=None):
        """The id rules of a request's stop test in the form lmx_seq_set_stop takes them: (eos ids, keyword id lists) or None.  EOS ids: `eos_token_id` of
        generate(); keyword ids: `keyword_ids` of every KeywordsStoppingCriteria-like object among the stopping criteria (llava/mm_utils.py:83-100 builds
        them from the worker's "stop" string).  Rules the device cannot hold (more than 4 ids / keywords, a keyword longer than 8 ids) are left to the
        host alone — which evaluates every criterion on the ids it reads back in any case; the device rule only keeps ids past the stop from being made."""
        eos = sorted(int(e) for e in eos_set if e is not None and int(e) >= 0)
        kws: List[List[int]] = []
        for c in (criteria or []):
            for k in (getattr(c, "keyword_ids", None) or []):
                ids = [int(v) for v in (k.tolist() if hasattr(k, "tolist") else list(k))]
                if ids:
                    kws.append(ids)
        if len(eos) > 4:
            eos = []
        if len(kws) > 4 or any(len(k) > 8 for k in kws):
            kws = []
        return (eos, kws) if (eos or kws) else None

    def _set_stop(self, seq, spec) -> None:
        if not spec:
            return
        eos, kws = spec
        e = (ctypes.c_int64 * max(1, len(eos)))(*eos)
        flat = [v for k in kws for v in k]
        f = (ctypes.c_int64 * max(1, len(flat)))(*flat)
        ln = (ctypes.c_int32 * max(1, len(kws)))(*[len(k) for k in kws])
        check(lib.lmx_seq_set_stop(seq, e, len(eos), f, ln, len(kws), stream_handle()), "lmx_seq_set_stop")

    def _prepare_request(self, ids, images, attention_mask, sampling, stop=None, feats=None) -> dict:
        """The RANK-LOCAL half of a request's prefill: image encode (the tower is replicated here; `feats` = the (features, hashes) pair of a batch-wide
        sharded pass, _preencode_requests, replaces it), splice, selection of the valid rows, a fresh sequence with its
        sampling state.  Nothing in here carries a decoder collective, so under tensor parallelism a failure (a bad request, an allocation) can still be
        agreed on by all ranks before the collective-bearing half runs (tp_serving.prefill_symmetric).  Returns {cache, embeds, valid}."""
        cache = None
        try:
            self._tls.plan_mask = None
            self._tls.plan_src = None
            self._tls.feats_override = feats
            try:
                _, _, mask, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(ids, None, attention_mask, None, None, images)
            finally:
                self._tls.feats_override = None
            if embeds is None:
                embeds = self.get_model().embed_tokens(ids.to(self.device))
                valid = None if attention_mask is None else attention_mask.bool()
            else:
                valid = self._tls.plan_mask if mask is None else mask.bool()
            # prefix reuse (reuse.PrefixCache): a finished request's sequence whose first rows are this request's first rows
            keys, reused = None, 0
            if self._prefix is not None and embeds.shape[0] == 1 and (valid is None or bool(valid.all())):
                keys = self._request_row_keys(ids, embeds.shape[1])
            if keys is not None:
                holder, n_common = self._prefix.take(keys)
                if holder is not None:
                    p = min(n_common, embeds.shape[1] - 1)        # at least one row goes through the decoder: its logits pick the first token
                    p -= p % 8                                     # the fused q|k|v epilogue wants pos0 % 8 == 0
                    if p >= self._prefix.min_rows and lib.lmx_seq_truncate(holder.seqs[0], p, stream_handle()) == 0:
                        cache, reused = holder, p
                        self._prefix.rows_reused += p
                    else:
                        holder.close()
            if cache is None:
                cache = LmxKVCache(self, 1)
            cache.row_keys, cache.reused_rows, cache.out_ref = keys, reused, None
            if reused:
                embeds = embeds[:, reused:]
                valid = None
            if sampling is not None:
                # the draw happens on the device (csrc/sampling.hip): temperature -> top-k -> top-p -> multinomial, keyed by a seed
                # taken from torch's CPU generator (so torch.manual_seed makes a request reproducible)
                temperature, top_p, top_k, seed = sampling
                check(lib.lmx_seq_set_sampling(cache.seqs[0], float(temperature), float(top_p if top_p is not None else 1.0), int(top_k or 0), int(seed)),
                      "lmx_seq_set_sampling")
            self._set_stop(cache.seqs[0], stop)          # before the prefill: its own pick (token 1) is tested too
            return {"cache": cache, "embeds": embeds, "valid": valid}
        except BaseException:
            if cache is not None:
                cache.close()
            raise

    def _request_row_keys(self, ids: torch.Tensor, T: int):
        """reuse.row_keys of the request just spliced on this thread (or of a text-only request), None if it cannot be keyed."""
        from .reuse import row_keys
        src = getattr(self._tls, "plan_src", None)
        if src is None:                                   # no image: the rows are the token ids
            k = ids.detach().cpu().numpy().astype(np.int64).reshape(-1)
            return k if len(k) == T and (k >= 0).all() else None
        hashes = getattr(self._tls, "image_hashes", None)
        if hashes is None or src.shape[0] != 1 or src.shape[1] != T:
            return None
        return row_keys(src[0], hashes, int(self._tls.plan_P))

    def _release_request_cache(self, cache: "LmxKVCache", out: Optional[Sequence[int]] = None) -> None:
        """End of a request: its sequence goes to the prefix cache — keyed by the prompt rows and the generated ids that have KV rows (all but the last
        one) — or back to the engine's pool."""
        keys = getattr(cache, "row_keys", None)
        if out is None:
            out = getattr(cache, "out_ref", None)
        pc = self._prefix
        if keys is None or pc is None or not cache.seqs:
            cache.close()
            return
        try:
            gen = np.asarray(list(out or [])[:-1], dtype=np.int64)
            full = np.concatenate([keys, gen]) if len(gen) else keys
            n = min(len(full), int(lib.lmx_seq_length(cache.seqs[0])))
            if n < pc.min_rows:
                cache.close()
                return
            cache.row_keys = None
            pc.put(full[:n], cache)
        except BaseException:  # noqa: BLE001
            cache.close()
            raise

    def _run_prepared(self, prepared: Sequence[dict], prefill_chunk: int = 0, return_logits: bool = False):
        """The collective-bearing half: the decoder prefill of prepared requests — one goes through lmx_prefill (same bits as generate()), several TOGETHER
        through lmx_prefill_batch (one GEMM per linear over the packed rows of all of them).  Every rank of a tensor-parallel group must call this with the
        same requests in the same order.  On failure the caches are the caller's to close."""
        if len(prepared) == 1:
            p = prepared[0]
            last = self._prefill_rows(p["cache"], p["embeds"], p["valid"], want_all=False, greedy=True, chunk=prefill_chunk)
            return last if return_logits else None
        packed = []
        for p in prepared:
            e, valid = p["embeds"][0], p["valid"]
            if valid is not None and not bool(valid[0].all()):
                e = e.index_select(0, torch.nonzero(valid[0].to(self.device), as_tuple=False).flatten())
            packed.append(e.to(self.dtype).contiguous())
        n = len(prepared)
        seqs = [p["cache"].seqs[0] for p in prepared]
        arr = (ctypes.c_void_p * n)(*[q.value if isinstance(q, ctypes.c_void_p) else q for q in seqs])
        eptr = (ctypes.c_void_p * n)(*[e.data_ptr() for e in packed])
        cnt = (ctypes.c_int32 * n)(*[int(e.shape[0]) for e in packed])
        check(lib.lmx_prefill_batch(self._h, arr, n, eptr, cnt, int(prefill_chunk) * n if prefill_chunk else 0, 1, stream_handle()), "lmx_prefill_batch")
        return None

    def _prefill_request(self, ids, images, attention_mask, sampling, prefill_chunk: int = 0, return_logits: bool = False, stop=None):
        """Image encode + splice + prefill of ONE request (ids [1, L]) into a fresh sequence; the first pick (argmax, or a draw when
        `sampling` = (temperature, top_p, top_k, seed) is given) is on the device when the stream gets there.  Shared by the request
        thread (generate) and the scheduler threads (batching.py, tp_serving.py)."""
        p = self._prepare_request(ids, images, attention_mask, sampling, stop)
        try:
            last = self._run_prepared([p], prefill_chunk, return_logits=return_logits)
            return (p["cache"], last) if return_logits else p["cache"]
        except BaseException:
            p["cache"].close()
            raise

    def _prefill_requests(self, reqs: Sequence[dict], prefill_chunk: int = 0) -> List["LmxKVCache"]:
        """Several requests prefilled TOGETHER (lmx_prefill_batch: one GEMM per linear over the packed rows of all of them).  reqs: dicts with ids [1, L],
        images, attention_mask, sampling (as _prefill_request takes them).  One request takes the single-sequence path (same bits as generate())."""
        prepared: List[dict] = []
        try:
            for r in reqs:
                prepared.append(self._prepare_request(r["ids"].to(self.device), r["images"], r["attention_mask"], r["sampling"], r.get("stop")))
            self._run_prepared(prepared, prefill_chunk)
            return [p["cache"] for p in prepared]
        except BaseException:
            for p in prepared:
                p["cache"].close()
            raise

    def _generate_one(self, ids, images, attention_mask, greedy, temperature, top_p, top_k, max_new_tokens, eos_set, streamer,
                      stopping_criteria, run_ahead, prefill_chunk) -> List[int]:
        if max_new_tokens <= 0:
            return []
        crit = list(stopping_criteria) if stopping_criteria is not None else []
        out: List[int] = []

        def make_emit(budget: int):
            def emit(tok: int) -> bool:
                out.append(tok)
                if streamer is not None:
                    streamer.put(torch.tensor([tok]))
                if tok in eos_set:
                    return True
                if crit:
                    full = torch.cat([ids.cpu(), torch.tensor([out], dtype=torch.long)], dim=1)
                    if any(c(full, None) for c in crit):
                        return True
                return len(out) >= budget
            return emit

        stop = self._stop_spec(eos_set, crit)           # the id rules go with the sequence to the device (lmx_seq_set_stop)
        batcher = self._batcher
        if batcher is not None and not prefill_chunk:
            prefill_chunk = self._batch_prefill_chunk
        if batcher is not None and (batcher.channel is not None or batcher.scheduler_prefill):
            # the scheduler thread prefills: under tensor parallelism because every call that carries a collective must be issued in an order the
            # leader broadcasts to the followers first (tp_serving.py); on one GPU (packed_prefill) because the requests waiting at that moment are
            # prefilled TOGETHER, one GEMM per linear over all their rows (lmx_prefill_batch).  The request thread hands the request over and waits
            sampling = None if greedy else (float(temperature), top_p, top_k, int(torch.randint(0, 2 ** 62, (1,)).item()))
            request = {"ids": ids.cpu(), "images": images, "attention_mask": None if attention_mask is None else attention_mask.cpu(),
                       "sampling": sampling, "prefill_chunk": int(prefill_chunk), "stop": stop, "out_ref": out}
            if getattr(batcher, "prepare_on_request_thread", False) and batcher.queued() < batcher.capacity:      # (a prepared request holds a sequence: bounded)
                # the rank-local half of the prefill here, on this request's own thread and stream (beside the running decode steps and the other requests'
                # prefills): image encode or feature-cache hit, splice, prefix-cache take.  The prefill thread then packs ready rows only
                prepared = self._prepare_request(ids.to(self.device), images, attention_mask, sampling, stop)
                request["prepared"], request["rows"] = prepared, int(prepared["embeds"].shape[1])
                # (from here the scheduler owns the prepared sequence: it closes it if the request never reaches a prefill)
            batcher.submit_request(request, make_emit, int(max_new_tokens))
            return out
        # With the batching scheduler on, image encode + prefill of concurrent requests run one at a time: k prefills sharing the GPU
        # all finish late (time to first token = k x one prefill for everybody), one after the other finishes the first after one.
        gate = self._prefill_gate if batcher is not None else None
        if gate is not None:
            gate.acquire()
        try:
            sampling = None if greedy else (float(temperature), top_p, top_k, self._draw_seed())
            cache = self._prefill_request(ids, images, attention_mask, sampling, prefill_chunk, stop=stop)
            seq = cache.seqs[0]
            if gate is not None:
                torch.cuda.current_stream(self.device).synchronize()      # the next request's prefill starts when this one is done
        finally:
            if gate is not None:
                gate.release()
        try:
            n_ctx = lib.lmx_seq_length(seq)
            budget = min(max_new_tokens, self.s_max - n_ctx)
            # A streamer wants every token as it is made.  Stopping criteria alone do not: they are evaluated per prefix on the ids read back in
            # run-ahead batches (same stop position as the reference's per-token check, mm_utils.py:94-107).  The id rules (EOS, keyword ids) are also
            # applied on the device by the pick itself, so no id exists past such a stop; only a TEXT-rule stop leaves ids behind it to drop
            interactive = streamer is not None
            emit = make_emit(budget)

            if batcher is not None:
                # continuous batching: this request's decode steps share the weight stream with every other live request
                host1 = (ctypes.c_int64 * 1)(); n1 = ctypes.c_int32(0)
                check(lib.lmx_seq_read_tokens(seq, host1, 1, ctypes.byref(n1), stream_handle()), "read_tokens")
                if not emit(int(host1[0])):
                    batcher.submit(seq, emit, room=budget - 1)
                return out
            host = (ctypes.c_int64 * (budget + 1))()
            n = ctypes.c_int32(0)
            done, consumed = False, 0
            # token 1 is the prefill's pick (argmax or draw); afterwards chain `ahead` steps on the device per host round trip.
            # Tokens produced past a stop are discarded (the cache is dropped with the sequence).
            produced = 1
            while True:
                check(lib.lmx_seq_read_tokens(seq, host, budget + 1, ctypes.byref(n), stream_handle()), "read_tokens")
                while consumed < min(n.value, produced) and not done:
                    done = emit(int(host[consumed])); consumed += 1
                if done or produced >= budget or n.value < produced:        # n < produced: the device-side rule stopped the sequence
                    break
                ahead = 1 if interactive else run_ahead
                ahead = max(1, min(ahead, budget - produced))
                check(lib.lmx_decode(self._h, seq, -1, ahead, None, 1, stream_handle()), "lmx_decode")
                produced += ahead
            return out
        finally:
            self._release_request_cache(cache, out)      # to the prefix cache (enable_reuse) or back to the pool


def _rope_theta(config) -> float:
    t = getattr(config, "rope_theta", None)
    if t is None:
        rp = getattr(config, "rope_parameters", None) or {}
        t = rp.get("rope_theta", 10000.0) if isinstance(rp, dict) else 10000.0
    return float(t)
