"""ctypes binding of libllava_mi355x.so (the C ABI in include/llava_mi355x.h).

The HIP extension is the product: there is NO CPU fallback here.  Importing this module loads the shared library
and raises ImportError with build instructions when it is missing (`python __graft_entry__.py` / `make -C csrc`).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libllava_mi355x.so")

LMX_ABI_VERSION = 1
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF, ACT_SILU_MUL = 0, 1, 2, 3
PROJ_LINEAR, PROJ_MLP_GELU, PROJ_IDENTITY = 0, 1, 2
FEATURE_PATCH, FEATURE_CLS_PATCH = 0, 1


class LmxError(RuntimeError):
    """Raised for every non-zero status of the native library (message from lmx_last_error())."""


class LmxConfig(Structure):
    _fields_ = [
        ("abi_version", c_int32), ("dtype", c_int32),
        ("hidden_size", c_int32), ("intermediate_size", c_int32), ("n_layers", c_int32), ("n_heads", c_int32),
        ("n_kv_heads", c_int32), ("head_dim", c_int32), ("vocab_size", c_int32),
        ("rms_eps", c_float), ("rope_theta", c_float), ("max_position", c_int32),
        ("v_hidden", c_int32), ("v_intermediate", c_int32), ("v_layers", c_int32), ("v_heads", c_int32),
        ("v_image_size", c_int32), ("v_patch_size", c_int32), ("v_ln_eps", c_float),
        ("select_layer", c_int32), ("select_feature", c_int32), ("projector_type", c_int32), ("projector_depth", c_int32),
        ("tp_rank", c_int32), ("tp_world", c_int32), ("gemm_variant", c_int32), ("reserved", c_int32 * 8),
    ]


def _load():
    # torch first: it preloads its bundled libamdhip64 / librccl by absolute path.  Our library must bind to THAT runtime
    # instance (same SONAMEs) — we share device pointers and streams with torch; loading the system copy first would put two
    # HIP runtimes in one process ("no ROCm-capable device is detected" from the second one).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built. Run `python __graft_entry__.py` (build()) "
            f"or `make -C llava-plus-codebase_amd/csrc`. There is no CPU fallback for this path.")
    try:
        return ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"failed to load {LIB_PATH}: {e}") from e


lib = _load()

_i32p, _i64p, _u8p, _f32p = POINTER(c_int32), POINTER(c_int64), POINTER(c_uint8), POINTER(c_float)

_SIGS = {
    "lmx_last_error": (c_char_p, []),
    "lmx_abi_version": (c_int32, []),
    "lmx_create": (c_int32, [POINTER(LmxConfig), POINTER(c_void_p)]),
    "lmx_destroy": (c_int32, [c_void_p]),
    "lmx_load_weight": (c_int32, [c_void_p, c_char_p, c_void_p, c_int32, c_int32, _i64p, c_void_p]),
    "lmx_finalize_weights": (c_int32, [c_void_p]),
    "lmx_set_rope_table": (c_int32, [c_void_p, c_void_p, c_int32]),
    "lmx_tp_unique_id": (c_int32, [c_void_p]),
    "lmx_tp_init": (c_int32, [c_void_p, c_void_p]),
    "lmx_tp_comm_ranks": (c_int32, [c_void_p]),
    "lmx_tp_p2p_local_handle": (c_int32, [c_void_p, c_void_p]),
    "lmx_tp_p2p_connect": (c_int32, [c_void_p, c_void_p]),
    "lmx_tp_p2p_enable": (c_int32, [c_void_p, c_int32]),
    "lmx_tp_p2p_status": (c_int32, [c_void_p, c_void_p]),
    "lmx_op_allreduce": (c_int32, [c_void_p, c_void_p, ctypes.c_uint64, c_void_p]),
    "lmx_tp_set_allreduce_hook": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "lmx_encode_images": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "lmx_vision_tower": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "lmx_preprocess_coeffs": (c_int32, [c_int32, c_int32, c_int32, c_int32, _i32p, _i32p, c_int32]),
    "lmx_preprocess_image": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, _f32p, _f32p, c_void_p, c_void_p]),
    "lmx_tokens_per_image": (c_int32, [c_void_p]),
    "lmx_set_vocab_limit": (c_int32, [c_void_p, c_int32]),
    "lmx_splice_plan": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                  _i32p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmx_gather_embeds": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "lmx_seq_create": (c_int32, [c_void_p, POINTER(c_void_p)]),
    "lmx_seq_destroy": (c_int32, [c_void_p]),
    "lmx_seq_set_sampling": (c_int32, [c_void_p, c_float, c_float, c_int32, ctypes.c_uint64]),
    "lmx_seq_set_stop": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "lmx_seq_stopped": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "lmx_seq_reset": (c_int32, [c_void_p]),
    "lmx_seq_length": (c_int32, [c_void_p]),
    "lmx_prefill": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "lmx_prefill_hidden": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "lmx_prefill_outputs": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "lmx_decode": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p]),
    "lmx_batch_create": (c_int32, [c_void_p, c_int32, POINTER(c_void_p)]),
    "lmx_batch_destroy": (c_int32, [c_void_p]),
    "lmx_decode_batch": (c_int32, [c_void_p, c_void_p, POINTER(c_void_p), c_int32, _i64p, c_int32, c_void_p, c_int32, _i64p, c_void_p]),
    "lmx_decode_batch_async": (c_int32, [c_void_p, c_void_p, POINTER(c_void_p), c_int32, c_int32, c_void_p, c_void_p]),
    "lmx_seq_read_tokens": (c_int32, [c_void_p, c_void_p, c_int32, _i32p, c_void_p]),
    "lmx_profile_enable": (c_int32, [c_void_p, c_int32]),
    "lmx_profile_read": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, _i32p]),
    "lmx_model_set_option": (c_int32, [c_void_p, ctypes.c_char_p, c_int32]),
    "lmx_op_gemm": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 9 + [c_void_p]),
    "lmx_op_gemv": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float] + [c_int32] * 8 + [c_void_p]),
    "lmx_op_rmsnorm": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "lmx_op_layernorm": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "lmx_op_rope_kv": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "lmx_op_rope_kv_rows": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "lmx_op_gemm_qkv_rope": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "lmx_op_flash_attn": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 8 + [c_float, c_int32, c_void_p]),
    "lmx_op_flash_attn_lse": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 8 + [c_float, c_int32, c_void_p, c_int32, c_void_p]),
    "lmx_op_attn_bwd_lse": (c_int32, [c_int32, c_int32] + [c_void_p] * 6 + [c_int32] + [c_void_p] * 3 + [c_int32] * 7 + [c_float, c_void_p]),
    "lmx_op_decode_attn": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 10 + [c_float, c_void_p, c_void_p]),
    "lmx_op_decode_fused": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                      c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "lmx_op_decode_attn_step": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmx_op_decode_kv_attn": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, ctypes.c_uint32, c_void_p, c_void_p, c_void_p]),
    "lmx_op_decode_attn_batch": (c_int32, [c_int32, c_int32, c_void_p, c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                           POINTER(c_void_p), c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int32, c_void_p]),
    "lmx_op_decode_attn_batch_tab_bytes": (c_size_t, [c_int32]),
    "lmx_op_decode_attn_ws_bytes": (c_size_t, [c_int32] * 4),
    "lmx_op_sample": (c_int32, [c_int32, c_void_p, c_int32, c_float, c_float, c_int32, ctypes.c_uint64, c_void_p, POINTER(ctypes.c_uint32), c_void_p, c_void_p, c_void_p]),
    "lmx_op_argmax": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "lmx_op_ce_loss": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p]),
    "lmx_op_rmsnorm_bwd": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "lmx_op_rmsnorm_bwd_add": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "lmx_op_swiglu_bwd": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "lmx_op_rope_bwd": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "lmx_op_transpose": (c_int32, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "lmx_op_gemm_wgrad": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "lmx_op_gemm_wgrad_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "lmx_op_elementwise": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "lmx_op_cast_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    "lmx_op_col_sum": (c_int32, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "lmx_op_gather_embed": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "lmx_op_embed_bwd": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "lmx_op_sumsq": (c_int32, [c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "lmx_prefill_batch": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "lmx_seq_copy": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "lmx_seq_truncate": (c_int32, [c_void_p, c_int32, c_void_p]),
    "lmx_op_hash128": (c_int32, [c_void_p, ctypes.c_uint64, c_int32, c_void_p, c_void_p]),
    "lmx_op_beam_topk": (c_int32, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "lmx_op_beam_sample_topk": (c_int32, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_float, ctypes.c_uint64, ctypes.c_uint32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lmx_op_adamw": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int32,
                               c_void_p, c_float, c_void_p]),
    "lmx_op_attn_bwd": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "lmx_op_im2col": (c_int32, [c_int32, c_void_p, c_void_p] + [c_int32] * 4 + [c_void_p]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)     # AttributeError here = header/library drift, fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)


def last_error() -> str:
    msg = lib.lmx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise LmxError(f"{what + ': ' if what else ''}{last_error()} (status {rc})")


def torch_dtype_code(dtype) -> int:
    import torch
    if dtype == torch.float32:
        return DTYPE_F32
    if dtype == torch.bfloat16:
        return DTYPE_BF16
    if dtype == torch.float16:
        return DTYPE_F16
    raise ValueError(f"unsupported dtype {dtype} (float32 | bfloat16 | float16)")


def ptr(t) -> c_void_p:
    """Raw device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream_handle() -> c_void_p:
    """hipStream_t of torch's current stream (PyTorch is only the container / stream provider)."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
