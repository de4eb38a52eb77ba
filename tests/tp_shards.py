"""TEST HELPER (not product code): tensor-parallel shard algebra of the decoder (Python statement of what Model::load_weight does in csrc/engine.cpp).

Megatron-style: q/k/v and gate/up are column-parallel (slices of output rows: heads / intermediate
columns), o_proj and down_proj are row-parallel (slices of input columns); lm_head is vocabulary-parallel (row slices, when the
vocabulary splits into 8-row-aligned shards); norms, embeddings, vision tower and projector are replicated.  Each rank's o_proj / down_proj output is a partial sum: all-reduce(sum) ×2 per layer, with
the residual added on rank 0's partial only.
"""
from __future__ import annotations

from typing import Tuple


def shard_slices(name: str, shape: Tuple[int, ...], n_heads: int, n_kv_heads: int, head_dim: int, intermediate: int,
                 rank: int, world: int):
    """Return (row_slice, col_slice) of tensor `name` owned by `rank`, or None when the tensor is replicated."""
    if world == 1:
        return None
    leaf = name.split(".", 3)[-1] if name.startswith("model.layers.") else name
    nh_l, nkv_l, i_l = n_heads // world, n_kv_heads // world, intermediate // world
    if leaf == "self_attn.q_proj.weight":
        return slice(rank * nh_l * head_dim, (rank + 1) * nh_l * head_dim), slice(None)
    if leaf in ("self_attn.k_proj.weight", "self_attn.v_proj.weight"):
        return slice(rank * nkv_l * head_dim, (rank + 1) * nkv_l * head_dim), slice(None)
    if leaf == "self_attn.o_proj.weight":
        return slice(None), slice(rank * nh_l * head_dim, (rank + 1) * nh_l * head_dim)
    if leaf in ("mlp.gate_proj.weight", "mlp.up_proj.weight"):
        return slice(rank * i_l, (rank + 1) * i_l), slice(None)
    if leaf == "mlp.down_proj.weight":
        return slice(None), slice(rank * i_l, (rank + 1) * i_l)
    if name == "lm_head.weight" and shape[0] % (8 * world) == 0:
        v_l = shape[0] // world
        return slice(rank * v_l, (rank + 1) * v_l), slice(None)
    return None


def shard_tensor(name, tensor, n_heads, n_kv_heads, head_dim, intermediate, rank, world):
    s = shard_slices(name, tuple(tensor.shape), n_heads, n_kv_heads, head_dim, intermediate, rank, world)
    return tensor if s is None else tensor[s[0], s[1]]
