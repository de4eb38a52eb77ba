"""One visual-instruction-tuning step on the MI355X kernels (BASELINE config 5).

What the reference runs per step (llava/train/train.py:805-1000 -> HF Trainer.training_step -> LlavaLlamaForCausalLM.forward with labels,
llava_llama.py:56-99; attention through llava/train/llama_flash_attn_monkey_patch.py:68-91; DeepSpeed ZeRO-2 per scripts/zero2.json:16-22;
AdamW, max_grad_norm 1.0 from the HF TrainingArguments defaults the launch scripts keep) and what stands here:

    forward     splice plan (the same lmx_splice_plan as inference: llava_arch.py:99-240) -> packed rows of all samples (the varlen form of
                flash_attn_unpadded_qkvpacked_func: no pad positions are computed) -> decoder layers on the forward kernels (GEMM, RMSNorm,
                RoPE + causal flash attention per sample, SiLU*mul) -> lm_head -> shifted cross-entropy with IGNORE_INDEX
    backward    csrc/train.hip: CE, RMSNorm, SwiGLU, RoPE, attention backward; dgrad / wgrad on the forward GEMM kernels (operands transposed
                once); embedding scatter-add; projector bias / GELU gradients.  The CLIP tower is frozen (clip_encoder.py:25): its features
                come in as data.
    optimiser   global grad-norm clip + AdamW on fp32 master weights and moments (csrc/train.hip adamw_kernel)
    ZeRO-2      ZeroPartition: parameters and gradients live in ONE flat buffer each, laid out in backward order and cut into buckets;
                a bucket whose gradients are complete is reduce-scattered while the backward pass continues (rank r keeps slice r of every
                bucket and the optimiser state for it only), updated slices are all-gathered back.  Collectives are torch.distributed
                (backend "nccl" = RCCL on the GPUs, one process per GPU); with a gloo group (CPU tests, single-GPU dry runs) device tensors
                are staged through the host.

This is the parity-first version: fp32-engine runs match torch autograd + torch.optim.AdamW on the oracle (tests/test_train_step_gpu.py).
`checkpoint=True` keeps only each layer's input and recomputes the layer in the backward pass (the reference's gradient_checkpointing).
`keep_layers` / `keep_budget_bytes` (round 6): with 288 GB per GPU the LAST k layers keep their activations anyway — as many as the budget holds — and are
not recomputed: every kept layer takes one forward (a quarter of its GEMM FLOPs) out of the step; gradients are the same bits either way.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _C, ops
from ._C import check, lib
from .constants import IGNORE_INDEX


def _round_up(n: int, m: int) -> int:
    return -(-n // m) * m


# ---------------------------------------------------------------------------------------------------------------------------------------
# ZeRO-2 bookkeeping + collectives (no kernels: runs on any device, covered by a gloo world_size-2 test on CPU)
# ---------------------------------------------------------------------------------------------------------------------------------------
class ZeroPartition:
    """Flat layout of named parameters, cut into buckets; rank r owns slice r of every bucket.

    sizes: [(name, numel)] in the order gradients become final (backward order).  Every parameter starts at a multiple of `align`
    elements (16-byte rows for the GEMMs); a bucket closes at the first parameter boundary past `bucket_elems` and is padded so that it
    splits evenly over `world` ranks."""

    def __init__(self, sizes: Sequence[Tuple[str, int]], world: int = 1, rank: int = 0, bucket_elems: int = 1 << 24, align: int = 64,
                 no_close_after: Sequence[str] = ()):
        assert world >= 1 and 0 <= rank < world
        self.world, self.rank, self.align = int(world), int(rank), int(align)
        self.offset: Dict[str, int] = {}
        self.numel: Dict[str, int] = {}
        self.bucket_of: Dict[str, int] = {}
        self.buckets: List[Tuple[int, int]] = []
        self.members: List[List[str]] = []
        pos, start, names = 0, 0, []
        hold = set(no_close_after)
        quantum = self.world * self.align
        for name, n in sizes:
            self.offset[name], self.numel[name] = pos, int(n)
            self.bucket_of[name] = len(self.buckets)
            names.append(name)
            pos = _round_up(pos + int(n), self.align)
            # a bucket never closes after a name in `no_close_after`: its padding (a multiple of world * align elements, not of the next tensor's
            # row size for world = 3, 5, 6, 7) would separate tensors the step uses as ONE operand (q | k | v)
            if pos - start >= bucket_elems and name not in hold:
                pos = start + _round_up(pos - start, quantum)
                self.buckets.append((start, pos)); self.members.append(names)
                start, names = pos, []
        if names:
            pos = start + _round_up(pos - start, quantum)
            self.buckets.append((start, pos)); self.members.append(names)
        self.total = pos
        self.shard_offset = []                      # where bucket b's owned slice sits in this rank's shard buffers
        acc = 0
        for s, e in self.buckets:
            self.shard_offset.append(acc)
            acc += (e - s) // self.world
        self.shard_elems = acc

    def owned(self, b: int) -> Tuple[int, int]:
        s, e = self.buckets[b]
        n = (e - s) // self.world
        return s + self.rank * n, s + (self.rank + 1) * n

    def shard_view(self, shard: torch.Tensor, b: int) -> torch.Tensor:
        s, e = self.buckets[b]
        n = (e - s) // self.world
        return shard[self.shard_offset[b]: self.shard_offset[b] + n]

    # ---- collectives ----------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _host_staged(group) -> bool:
        import torch.distributed as dist
        return dist.get_backend(group) == "gloo"

    def reduce_scatter(self, flat_g: torch.Tensor, g_shard: torch.Tensor, b: int, group=None, async_op: bool = False):
        """g_shard's slice for bucket b := sum over ranks of slice `rank` of bucket b of flat_g.  Returns a work handle or None."""
        out = self.shard_view(g_shard, b)
        s, e = self.buckets[b]
        if self.world == 1:
            if out.data_ptr() != flat_g[s:e].data_ptr():          # one rank: the shard buffer may BE the flat buffer (TrainStep does that): nothing to move
                out.copy_(flat_g[s:e])
            return None
        import torch.distributed as dist
        if self._host_staged(group):
            # gloo has no reduce-scatter: all-reduce the bucket on the host and keep the owned slice
            t = flat_g[s:e].float().cpu()
            dist.all_reduce(t, group=group)
            o0, o1 = self.owned(b)
            out.copy_(t[o0 - s: o1 - s].to(out.dtype))
            return None
        return dist.reduce_scatter_tensor(out, flat_g[s:e], group=group, async_op=async_op)

    def all_gather(self, flat_p: torch.Tensor, p_shard: torch.Tensor, b: int, group=None, async_op: bool = False):
        """bucket b of flat_p := concatenation over ranks of their p_shard slices."""
        src = self.shard_view(p_shard, b)
        s, e = self.buckets[b]
        if self.world == 1:
            if src.data_ptr() != flat_p[s:e].data_ptr():
                flat_p[s:e].copy_(src)
            return None
        import torch.distributed as dist
        if self._host_staged(group):
            parts = [torch.empty(src.shape, dtype=src.dtype) for _ in range(self.world)]
            dist.all_gather(parts, src.detach().cpu().contiguous(), group=group)
            flat_p[s:e].copy_(torch.cat(parts).to(flat_p.device))
            return None
        return dist.all_gather_into_tensor(flat_p[s:e], src, group=group, async_op=async_op)

    def all_reduce_scalar(self, x: torch.Tensor, group=None) -> None:
        if self.world == 1:
            return
        import torch.distributed as dist
        if self._host_staged(group) and x.is_cuda:
            t = x.cpu(); dist.all_reduce(t, group=group); x.copy_(t)
        else:
            dist.all_reduce(x, group=group)


# ---------------------------------------------------------------------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------------------------------------------------------------------
class TrainStep:
    """Trainable LLM + mm_projector of one LLaVA model on one rank of a data-parallel group.

    config: the LlavaConfig fields the inference model reads (hidden_size, intermediate_size, num_hidden_layers, num_attention_heads,
    num_key_value_heads, rms_norm_eps, vocab_size, rope theta, mm_projector_type, mm_hidden_size).  weights: HF-named tensors of the LLM
    and the projector (`model.mm_projector.*` or `mm_projector.*`)."""

    def __init__(self, config, weights: Mapping[str, torch.Tensor], dtype=torch.bfloat16, device="cuda", lr=2e-5, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, max_grad_norm=1.0, group=None, bucket_elems: Optional[int] = None, checkpoint: bool = False, max_positions: int = 2048,
                 keep_layers: int = 0, keep_budget_bytes: Optional[int] = None, direct_wgrad: bool = True, attn_streams: int = 4):
        from .model import _projector_kind, _rope_theta
        self.config, self.dtype, self.device = config, dtype, torch.device(device)
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = float(lr), tuple(betas), float(eps), float(weight_decay), float(max_grad_norm)
        self.checkpoint = bool(checkpoint)
        self.keep_layers, self.keep_budget_bytes = int(keep_layers), keep_budget_bytes
        self.last_kept_layers = 0
        self._kv_pairs: Dict = {}
        self.attn_streams = int(attn_streams)    # side streams the per-sample attention launches of a layer spread over (0 / 1: the step's stream only)
        self._side: List = []
        self.direct_wgrad = bool(direct_wgrad)   # weight gradients straight from dy / x (csrc/gemm8t.hip) where the shapes allow; False: the two-transpose path
        self.group = group
        self.world, self.rank = 1, 0
        if group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        c = config
        self.H, self.I, self.L, self.V = c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.vocab_size
        self.nh = c.num_attention_heads
        self.nkv = getattr(c, "num_key_value_heads", None) or self.nh
        self.D = self.H // self.nh
        self.rms_eps = float(c.rms_norm_eps)
        self.kmult = 64 if dtype in (torch.bfloat16, torch.float16) else 16          # GEMM contraction quantum (csrc/gemm.hip)
        assert self.H % 64 == 0 and self.I % 64 == 0 and self.V % 8 == 0, "hidden / intermediate sizes must be multiples of 64"
        kind, depth = _projector_kind(getattr(c, "mm_projector_type", "linear"))
        self.proj_names: List[str] = []                                                # identity: nothing to train
        if kind == _C.PROJ_LINEAR:
            self.proj_names = ["mm_projector.weight", "mm_projector.bias"]
        elif kind == _C.PROJ_MLP_GELU:
            for j in range(depth):
                self.proj_names += [f"mm_projector.{2 * j}.weight", f"mm_projector.{2 * j}.bias"]
        w = {}
        for k, v in weights.items():
            k = k.replace("model.mm_projector.", "mm_projector.")
            w[k] = v
        # ---- flat layout, backward order ---------------------------------------------------------------------------------------------
        order: List[str] = ["lm_head.weight", "model.norm.weight"]
        for l in reversed(range(self.L)):
            p = f"model.layers.{l}."
            order += [p + "mlp.down_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", p + "post_attention_layernorm.weight",
                      p + "self_attn.o_proj.weight", p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                      p + "input_layernorm.weight"]
        for j in reversed(range(len(self.proj_names) // 2)):
            order += [self.proj_names[2 * j + 1], self.proj_names[2 * j]]
        order.append("model.embed_tokens.weight")
        missing = [n for n in order if n not in w]
        if missing:
            raise KeyError(f"weights missing for the trainable part: {missing[:4]}...")
        self.shapes = {n: tuple(w[n].shape) for n in order}
        if bucket_elems is None:
            bucket_elems = self.H * self.H                       # DeepSpeed's "auto" reduce_bucket_size = hidden_size^2
        # q | k | v of a layer form one fused operand of the step: no bucket boundary (padding) between them, whatever the world size
        fused = [f"model.layers.{l}.self_attn.{p}_proj.weight" for l in range(self.L) for p in ("q", "k")]
        self.part = ZeroPartition([(n, int(np.prod(self.shapes[n]))) for n in order], self.world, self.rank, bucket_elems, no_close_after=fused)
        P = self.part
        self.flat_p = torch.zeros(P.total, dtype=dtype, device=self.device)
        self.flat_g = torch.zeros(P.total, dtype=dtype, device=self.device)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for n in order:
            o, k = P.offset[n], P.numel[n]
            self.p[n] = self.flat_p[o:o + k].view(self.shapes[n])
            self.g[n] = self.flat_g[o:o + k].view(self.shapes[n])
            self.p[n].copy_(w[n].to(device=self.device, dtype=dtype))
        # q | k | v rows are adjacent in the flat buffer: one fused [(nh + 2 nkv) D, H] operand without a copy
        self.qkv_w, self.qkv_g = [], []
        nq = (self.nh + 2 * self.nkv) * self.D
        for l in range(self.L):
            o = P.offset[f"model.layers.{l}.self_attn.q_proj.weight"]
            assert P.offset[f"model.layers.{l}.self_attn.k_proj.weight"] == o + self.nh * self.D * self.H
            assert P.offset[f"model.layers.{l}.self_attn.v_proj.weight"] == o + (self.nh + self.nkv) * self.D * self.H
            self.qkv_w.append(self.flat_p[o:o + nq * self.H].view(nq, self.H))
            self.qkv_g.append(self.flat_g[o:o + nq * self.H].view(nq, self.H))
        # ---- optimiser state for the owned slices only (ZeRO-2) -----------------------------------------------------------------------
        n_own = P.shard_elems
        self.master = torch.empty(n_own, dtype=torch.float32, device=self.device)
        for b in range(len(P.buckets)):
            o0, o1 = P.owned(b)
            P.shard_view(self.master, b).copy_(self.flat_p[o0:o1].float())
        self.exp_avg = torch.zeros(n_own, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(n_own, dtype=torch.float32, device=self.device)
        if self.world == 1:
            # one rank owns every bucket whole and the shard layout IS the flat layout: the "shards" are the flat buffers themselves — no reduce-scatter / all-gather
            # stand-in copies (2 x 13.5 GB moved per 7B step) and 27 GB less resident, which the kept-activation budget turns into five more layers
            assert n_own == P.total and all(P.shard_offset[b] == P.buckets[b][0] for b in range(len(P.buckets)))
            self.g_shard, self.p_shard = self.flat_g, self.flat_p
        else:
            self.g_shard = torch.zeros(n_own, dtype=dtype, device=self.device)
            self.p_shard = torch.zeros(n_own, dtype=dtype, device=self.device)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.step_count = 0
        self.last_grad_norm: Optional[float] = None
        # ---- RoPE table, exactly as the inference model builds it ---------------------------------------------------------------------
        theta = _rope_theta(c)
        inv_freq = 1.0 / (theta ** (torch.arange(0, self.D, 2, dtype=torch.int64).to(torch.float32) / self.D))
        freqs = torch.arange(int(max_positions), dtype=torch.float32)[:, None] * inv_freq[None, :]
        self.rope = torch.cat([freqs.cos(), freqs.sin()], dim=-1).contiguous().to(self.device)
        self.max_positions = int(max_positions)
        self._pending: List = []
        self._ready: Dict[int, int] = {}
        self.frozen: set = set()
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.world > 1 and self.device.type == "cuda") else None

    # ---- helpers ------------------------------------------------------------------------------------------------------------------------
    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor) -> None:
        """out [N, K] = dy[M, N]^T @ x[M, K]   (M is a multiple of the contraction quantum: the packed rows are padded to it)."""
        if self.direct_wgrad and ops.wgrad_direct_ok(dy, x, out):
            ops.gemm_wgrad(dy, x, out)                               # both operands in their forward layout (csrc/gemm8t.hip): no transposed copies
        else:
            ops.gemm(ops.transpose_padded(dy, self.kmult), ops.transpose_padded(x, self.kmult), out=out)

    def _dgrad(self, dy: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dy [M, N] @ w [N, K] -> [M, K] (+ residual)."""
        N = w.shape[0]
        wt = ops.transpose_padded(w, self.kmult)                 # [K, Np]
        if wt.shape[1] != N:
            pad = torch.zeros((dy.shape[0], wt.shape[1]), dtype=dy.dtype, device=dy.device)
            pad[:, :N].copy_(dy)
            dy = pad
        return ops.gemm(dy, wt, residual=residual)

    def _attn_streams(self, n_items: int):
        """Context manager for the per-sample attention launches of one layer: `with self._attn_streams(n) as on: with on(i) as lane: ...` runs item i on side stream
        i % attn_streams (lane = that index; 0 streams or fewer than two items: everything stays on the step's stream, lane 0).  The side streams wait for the step's
        stream on entry and the step's stream waits for all of them on exit.  Why: a 2048-row sample's attention launches are 512 workgroups of very different length
        (1 .. 32 key tiles); run one after the other each launch ends with most CUs idle behind its heaviest workgroups, run side by side the next sample's workgroups
        fill them.  No arithmetic changes (no atomics anywhere: same bits)."""
        import contextlib
        step = self

        class _Ctx:
            def __enter__(self_c):
                self_c.use = step.attn_streams >= 2 and n_items >= 2 and step.device.type == "cuda"
                if self_c.use:
                    while len(step._side) < step.attn_streams:
                        step._side.append(torch.cuda.Stream(device=step.device))
                    self_c.main = torch.cuda.current_stream(step.device)
                    ev = torch.cuda.Event(); ev.record(self_c.main)
                    for s in step._side[:step.attn_streams]:
                        s.wait_event(ev)

                @contextlib.contextmanager
                def on(i):
                    if not self_c.use:
                        yield 0
                    else:
                        k = i % step.attn_streams
                        with torch.cuda.stream(step._side[k]):
                            yield k
                return on

            def __exit__(self_c, *exc):
                if self_c.use:
                    for s in step._side[:step.attn_streams]:
                        self_c.main.wait_stream(s)
                return False
        return _Ctx()

    def _kv_scratch(self, s_max: int, dtype, lane: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """K / V^T cache of one sample for the forward attention: a scratch pair per length, zeroed ONCE and reused by every span, layer and step (launches on one
        stream: the next rope_kv overwrites it after the previous flash attention has read it).  Slots past a sample's length hold earlier samples' finite values;
        the kernel masks them.  Allocating zeroed caches per span was 0.5 GB of fills per layer at 16 x 2048."""
        key = (int(s_max), dtype, int(lane))                           # one pair per side stream: samples on different streams run at the same time
        if key not in self._kv_pairs:
            self._kv_pairs[key] = ops.alloc_kv(self.nkv, int(s_max), self.D, dtype, self.device)
        return self._kv_pairs[key]

    @staticmethod
    def _zero_rows_outside(t: torch.Tensor, spans: List[Tuple[int, int]]) -> None:
        """Zero the rows of `t` that no span covers (the padding of the packed rows): the spans' rows are about to be overwritten, a fill of the whole
        buffer (268 - 805 MB per layer at 16 x 2048) is not needed."""
        at = 0
        for a, b in sorted(spans):
            if a > at:
                t[at:a].zero_()
            at = max(at, b)
        if at < t.shape[0]:
            t[at:].zero_()

    def layer_activation_bytes(self, rows: int) -> int:
        """Bytes one decoder layer's stash holds beyond its input rows (what _layer_forward returns: h, q|k|v with q and k rotated, attn, x1, h2, gate, up, act)."""
        es = torch.empty((), dtype=self.dtype).element_size()
        per_row = 3 * self.H + (self.nh + 2 * self.nkv) * self.D + self.nh * self.D + 3 * self.I
        return int(rows) * per_row * es

    def set_trainable(self, predicate) -> None:
        """Freeze every parameter whose name (as in self.p: HF names, the projector as `mm_projector.*`) the predicate rejects — the reference's
        `requires_grad_(False)` (llava/train/train.py:850-851, 925-934: freeze_backbone, tune_mm_mlp_adapter, freeze_mm_mlp_adapter).  A frozen tensor's
        gradient is cleared the moment it is final, before its bucket is reduced: it adds nothing to the clipping norm, its Adam moments stay zero and the
        update leaves its bits unchanged.  (Weight decay would still shrink it: refused.)"""
        frozen = {n for n in self.p if not predicate(n)}
        if frozen and self.wd != 0.0:
            raise ValueError("frozen parameters with weight_decay != 0 are not supported (the reference's scripts train with weight_decay 0.)")
        self.frozen = frozen

    def _mark_ready(self, names: Sequence[str]) -> None:
        """Gradients of `names` are final: reduce-scatter every bucket that just became complete, on the communication stream."""
        P = self.part
        for n in names:
            if n in self.frozen:
                self.g[n].zero_()
            b = P.bucket_of[n]
            self._ready[b] = self._ready.get(b, 0) + 1
            if self._ready[b] == len(P.members[b]):
                if self.comm_stream is not None:
                    self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self.comm_stream):
                        h = P.reduce_scatter(self.flat_g, self.g_shard, b, self.group, async_op=False)
                else:
                    h = P.reduce_scatter(self.flat_g, self.g_shard, b, self.group)
                self._pending.append(h)

    def _plan(self, input_ids, attention_mask, labels, slot_rows: np.ndarray):
        ids_h = np.ascontiguousarray(input_ids.detach().cpu().numpy().astype(np.int64))
        B, Lq = ids_h.shape
        mask_h = None if attention_mask is None else np.ascontiguousarray(attention_mask.detach().cpu().numpy().astype(np.uint8))
        lab_h = None if labels is None else np.ascontiguousarray(labels.detach().cpu().numpy().astype(np.int64))
        vp = lambda a: ctypes.c_void_p(0) if a is None else ctypes.c_void_p(a.ctypes.data)
        max_len = getattr(self.config, "tokenizer_model_max_length", None) or 0
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        T = ctypes.c_int32(0)
        args = (vp(ids_h), vp(mask_h), vp(lab_h), B, Lq, 0, vp(slot_rows), len(slot_rows), int(max_len), int(left))
        if lib.lmx_splice_plan(*args, ctypes.byref(T), None, None, None, None):
            raise IndexError(_C.last_error())
        Tn = T.value
        src = np.empty((B, Tn), np.int32); om = np.empty((B, Tn), np.uint8); op = np.empty((B, Tn), np.int64); ol = np.empty((B, Tn), np.int64)
        check(lib.lmx_splice_plan(*args, ctypes.byref(T), vp(src), vp(om), vp(op), vp(ol)), "lmx_splice_plan")
        return src, om.astype(bool), ol

    # ---- forward + backward ---------------------------------------------------------------------------------------------------------------
    def _layer_forward(self, l: int, x: torch.Tensor, spans: List[Tuple[int, int]]):
        p = f"model.layers.{l}."
        W, nh, nkv, D = self.p, self.nh, self.nkv, self.D
        h = ops.rmsnorm(x, W[p + "input_layernorm.weight"], self.rms_eps)
        qkv = ops.gemm(h, self.qkv_w[l])
        attn = torch.empty((x.shape[0], nh * D), dtype=x.dtype, device=x.device)
        self._zero_rows_outside(attn, spans)                                          # padding rows feed o_proj: finite zeros, as before
        # the forward's log-sum-exp, kept for the backward (FlashAttention-2's statistics); allocated on the step's stream BEFORE the samples spread over the side streams
        # (one allocation for the layer, no fill: rows past a sample's length are never written and only ever read by the key-side kernel's tile staging, which
        # discards them by SELECT, not by a multiplication — tests/test_wgrad_gpu.py runs the backward on a NaN-filled buffer)
        if x.dtype == torch.float32:
            lses = []
        else:
            tp = [_round_up(b - a, 64) for a, b in spans]
            flat = torch.empty(nh * sum(tp), dtype=torch.float32, device=x.device)
            lses, at = [], 0
            for n in tp:
                lses.append(flat[at:at + nh * n].view(nh, n)); at += nh * n
        with self._attn_streams(len(spans)) as on:
            for i, (a, b) in enumerate(spans):
                with on(i) as lane:
                    Tn = b - a
                    kc, vt = self._kv_scratch(_round_up(Tn, 128), x.dtype, lane)
                    rows = qkv[a:b]
                    ops.rope_kv(rows, kc, vt, self.rope, 0, nh, nkv, D, k_rows=True)   # q AND k rotated in place (the backward reads both from qkv), k / v into the caches
                    if x.dtype == torch.float32:  # fp32 verification mode: the VALU attention kernel, as the inference engine's fp32 prefill
                        ops.decode_attn(rows, kc, vt, Tn, 0, 0, nh, nkv, D, True, q_stride=qkv.stride(0), out=attn[a:b])
                    else:
                        ops.flash_attn(rows, kc, vt, Tn, Tn, 0, nh, nkv, D, True, q_stride=qkv.stride(0), out=attn[a:b], lse=lses[i])
        x1 = ops.gemm(attn, W[p + "self_attn.o_proj.weight"], residual=x)
        h2 = ops.rmsnorm(x1, W[p + "post_attention_layernorm.weight"], self.rms_eps)
        gate = ops.gemm(h2, W[p + "mlp.gate_proj.weight"])
        up = ops.gemm(h2, W[p + "mlp.up_proj.weight"])
        act = ops.elementwise(ops.EW_SWIGLU, gate, up)
        x2 = ops.gemm(act, W[p + "mlp.down_proj.weight"], residual=x1)
        return x2, dict(x=x, h=h, qkv=qkv, attn=attn, lse=lses, x1=x1, h2=h2, gate=gate, up=up, act=act)

    def _layer_backward(self, l: int, st: dict, d: torch.Tensor, spans: List[Tuple[int, int]]) -> torch.Tensor:
        p = f"model.layers.{l}."
        W, G, nh, nkv, D = self.p, self.g, self.nh, self.nkv, self.D
        dt = _C.torch_dtype_code(d.dtype)
        # MLP: x2 = x1 + down(silu(gate) * up)
        d_act = self._dgrad(d, W[p + "mlp.down_proj.weight"])
        self._wgrad(d, st["act"], G[p + "mlp.down_proj.weight"])
        dg, du = ops.swiglu_bwd(st["gate"], st["up"], d_act)
        self._wgrad(dg, st["h2"], G[p + "mlp.gate_proj.weight"])
        self._wgrad(du, st["h2"], G[p + "mlp.up_proj.weight"])
        dh2 = self._dgrad(du, W[p + "mlp.up_proj.weight"], residual=self._dgrad(dg, W[p + "mlp.gate_proj.weight"]))
        dx1, dw = ops.rmsnorm_bwd(st["x1"], W[p + "post_attention_layernorm.weight"], dh2, self.rms_eps, residual=d)     # d joins over the skip connection
        ops.cast_f32(dw, G[p + "post_attention_layernorm.weight"])
        # attention: x1 = x + o_proj(attn)
        d_attn = self._dgrad(dx1, W[p + "self_attn.o_proj.weight"])
        self._wgrad(dx1, st["attn"], G[p + "self_attn.o_proj.weight"])
        qkv = st["qkv"]
        dqkv = torch.empty_like(qkv)
        self._zero_rows_outside(dqkv, spans)
        ld = qkv.stride(0)
        es = qkv.element_size()
        ko, vo = nh * D * es, (nh + nkv) * D * es                                     # byte offsets of the k / v columns inside a q|k|v row
        with self._attn_streams(len(spans)) as on:
            for i, (a, b) in enumerate(spans):
                with on(i):
                    Tn = b - a
                    rows, dq_rows = qkv[a:b], dqkv[a:b]
                    # q, k (both rotated in place by the forward) and v are column windows of the q|k|v rows; dq, dk, dv go straight into the same windows of dqkv
                    at = lambda t, o: ctypes.c_void_p(t.data_ptr() + o)
                    if st["lse"]:          # 16-bit step: the forward's log-sum-exp and output come in (FlashAttention-2's form): no statistics sweep in the backward
                        lse, o_rows = st["lse"][i], st["attn"][a:b]
                        check(lib.lmx_op_attn_bwd_lse(dt, D, _C.ptr(rows), at(rows, ko), at(rows, vo), _C.ptr(o_rows), _C.ptr(d_attn[a:b]), _C.ptr(lse), lse.stride(0),
                                                      _C.ptr(dq_rows), at(dq_rows, ko), at(dq_rows, vo), Tn, nh, nkv, ld, ld, d_attn.stride(0), o_rows.stride(0),
                                                      1.0 / math.sqrt(D), _C.stream_handle()), "attn_bwd_lse")
                    else:                  # fp32 verification step: the two-pass VALU kernels recompute the statistics
                        s1 = torch.empty(Tn * nkv * D, dtype=torch.float32, device=d.device); s2 = torch.empty_like(s1)
                        check(lib.lmx_op_attn_bwd(dt, D, _C.ptr(rows), at(rows, ko), at(rows, vo), _C.ptr(d_attn[a:b]), _C.ptr(dq_rows), _C.ptr(s1), _C.ptr(s2),
                                                  at(dq_rows, ko), at(dq_rows, vo), Tn, nh, nkv, ld, ld, d_attn.stride(0), 1.0 / math.sqrt(D), _C.stream_handle()),
                              "attn_bwd")
                    # un-rotate dq | dk in place: they are adjacent heads of the same rows
                    check(lib.lmx_op_rope_bwd(dt, _C.ptr(dq_rows), _C.ptr(dq_rows), _C.ptr(self.rope), 0, Tn, nh + nkv, D, ld, _C.stream_handle()), "rope_bwd")
        dh = self._dgrad(dqkv, self.qkv_w[l])
        self._wgrad(dqkv, st["h"], self.qkv_g[l])
        dx, dw = ops.rmsnorm_bwd(st["x"], W[p + "input_layernorm.weight"], dh, self.rms_eps, residual=dx1)
        ops.cast_f32(dw, G[p + "input_layernorm.weight"])
        self._mark_ready([p + "mlp.down_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", p + "post_attention_layernorm.weight",
                          p + "self_attn.o_proj.weight", p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                          p + "input_layernorm.weight"])
        return dx

    def _projector_forward(self, f: torch.Tensor):
        """f [rows, mm_hidden] (frozen tower features, rows padded to the contraction quantum) -> (features [rows, H], stash)."""
        W = self.p
        if not self.proj_names:
            return f, None
        zs, ins = [], []
        x = f
        n_lin = len(self.proj_names) // 2
        for j in range(n_lin):
            if j > 0:
                zs.append(x)
                x = ops.elementwise(ops.EW_GELU, x)
            ins.append(x)
            x = ops.gemm(x, W[self.proj_names[2 * j]], bias=W[self.proj_names[2 * j + 1]])
        return x, dict(zs=zs, ins=ins)

    def _projector_backward(self, st: Optional[dict], d: torch.Tensor) -> None:
        if st is None:
            return
        W, G = self.p, self.g
        n_lin = len(self.proj_names) // 2
        for j in reversed(range(n_lin)):
            wn, bn = self.proj_names[2 * j], self.proj_names[2 * j + 1]
            self._wgrad(d, st["ins"][j], G[wn])
            ops.cast_f32(ops.col_sum(d), G[bn])
            if j > 0:
                da = self._dgrad(d, W[wn])
                d = ops.elementwise(ops.EW_GELU_BWD, st["zs"][j - 1], da)
        self._mark_ready(self.proj_names)

    def forward_backward(self, input_ids: torch.Tensor, labels: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                         image_features: Optional[torch.Tensor] = None, slot_rows: Optional[Sequence[int]] = None, backward: bool = True):
        """One micro-batch.  input_ids / labels / attention_mask: [B, L] (ids carry IMAGE_TOKEN_INDEX markers); image_features: output of the
        frozen tower, [n_images, P, mm_hidden] (or a flat [rows, mm_hidden] with slot_rows = rows per image slot).
        Returns (loss fp32 device scalar, number of counted label positions).  Gradients land in self.g (flat_g)."""
        dev, dt = self.device, self.dtype
        if image_features is not None and image_features.dim() == 3:
            slot_rows = [image_features.shape[1]] * image_features.shape[0]
            image_features = image_features.reshape(-1, image_features.shape[-1])
        slot_rows_np = np.ascontiguousarray(np.asarray(slot_rows if slot_rows is not None else [], dtype=np.int32))
        src, valid, lab = self._plan(input_ids, attention_mask, labels, slot_rows_np)
        B, Tn = src.shape
        # ---- pack the valid positions of all samples into one row block (varlen), padded with empty rows to the contraction quantum ----
        spans, rows_src, rows_lab, at = [], [], [], 0
        for b in range(B):
            idx = np.nonzero(valid[b])[0]
            if idx.size == 0:
                continue
            if idx.size > self.max_positions:
                raise ValueError("sequence longer than the RoPE table")
            s, l = src[b, idx], lab[b, idx].copy()
            l[0] = IGNORE_INDEX            # the row before a sample's first token belongs to the previous sample: it predicts nothing
            spans.append((at, at + idx.size)); at += idx.size
            rows_src.append(s); rows_lab.append(l)
        Np = _round_up(max(at, 1), self.kmult)
        src_p = np.full(Np, -1, np.int32); lab_p = np.full(Np, IGNORE_INDEX, np.int64)
        src_p[:at] = np.concatenate(rows_src); lab_p[:at] = np.concatenate(rows_lab)
        src_d = torch.from_numpy(src_p).to(dev); lab_d = torch.from_numpy(lab_p).to(dev)
        # ---- projector on the frozen tower's features, embedding gather + splice ---------------------------------------------------------
        feats, pst, f_rows = None, None, 0
        if image_features is not None:
            f_rows = image_features.shape[0]
            f = torch.zeros((_round_up(f_rows, self.kmult), image_features.shape[1]), dtype=dt, device=dev)
            f[:f_rows].copy_(image_features.to(device=dev, dtype=dt))
            feats, pst = self._projector_forward(f)
        x = ops.gather_embed(src_d, self.p["model.embed_tokens.weight"], feats)
        # ---- decoder ----------------------------------------------------------------------------------------------------------------------
        stash = []
        keep_from = 0
        if self.checkpoint:
            # activations of the LAST layers stay (they are consumed first by the backward): the explicit count, or as many as the byte budget holds
            kept = min(self.L, max(0, self.keep_layers))
            if self.keep_budget_bytes is not None:
                kept = max(kept, min(self.L, int(self.keep_budget_bytes // max(1, self.layer_activation_bytes(Np)))))
            keep_from = self.L - kept
            self.last_kept_layers = kept
        for l in range(self.L):
            x_in = x
            x, st = self._layer_forward(l, x, spans)
            stash.append({"x": x_in} if (self.checkpoint and l < keep_from) else st)
        hn = ops.rmsnorm(x, self.p["model.norm.weight"], self.rms_eps)
        logits = ops.gemm(hn, self.p["lm_head.weight"])
        loss, count, dlogits = ops.ce_loss(logits[None], lab_d[None], ignore_index=IGNORE_INDEX, want_grad=backward, grad=1.0 / self.world)
        if not backward:
            return loss, count
        # ---- backward (gradients of a data-parallel group are averaged: each rank back-propagates loss / world) ---------------------------
        self._ready.clear(); self._pending.clear()
        dlogits = dlogits[0]
        dhn = self._dgrad(dlogits, self.p["lm_head.weight"])
        self._wgrad(dlogits, hn, self.g["lm_head.weight"])
        d, dw = ops.rmsnorm_bwd(x, self.p["model.norm.weight"], dhn, self.rms_eps)
        ops.cast_f32(dw, self.g["model.norm.weight"])
        self._mark_ready(["lm_head.weight", "model.norm.weight"])
        del logits, dlogits
        for l in reversed(range(self.L)):
            st = stash[l]
            if len(st) == 1:                                     # only the layer's input was kept: recompute it
                _, st = self._layer_forward(l, st["x"], spans)
            d = self._layer_backward(l, st, d, spans)
            stash[l] = None
        dE = torch.zeros((self.V, self.H), dtype=torch.float32, device=dev)
        dfeats = torch.zeros_like(feats) if (feats is not None and pst is not None) else None
        ops.embed_bwd(src_d, d, dE, dfeats)
        if pst is not None:
            self._projector_backward(pst, dfeats)
        elif self.proj_names:
            for n in self.proj_names:          # a text-only micro-batch: the projector took no part
                self.g[n].zero_()
            self._mark_ready(self.proj_names)
        ops.cast_f32(dE.view(-1), self.g["model.embed_tokens.weight"].view(-1))
        self._mark_ready(["model.embed_tokens.weight"])
        return loss, count

    # ---- optimiser ------------------------------------------------------------------------------------------------------------------------
    def optimizer_step(self) -> None:
        """Clip by the global norm of the (averaged) gradients, AdamW on this rank's slices, all-gather the updated parameters."""
        P = self.part
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self.comm_stream is not None:
            cur.wait_stream(self.comm_stream)
        for h in self._pending:
            if h is not None:
                h.wait()
        self._pending.clear()
        self.step_count += 1
        self.gnorm_sq.zero_()
        ops.sumsq(self.g_shard, self.gnorm_sq)                   # padding elements of the shards are zero
        P.all_reduce_scalar(self.gnorm_sq, self.group)
        for b in range(len(P.buckets)):
            view = lambda t: P.shard_view(t, b)
            ops.adamw(view(self.p_shard), view(self.g_shard), view(self.master), view(self.exp_avg), view(self.exp_avg_sq), self.lr, self.betas, self.eps,
                      self.wd, self.step_count, gnorm_sq=self.gnorm_sq if self.max_grad_norm > 0 else None, max_grad_norm=self.max_grad_norm)
            P.all_gather(self.flat_p, self.p_shard, b, self.group)
        self.last_grad_norm = None

    def grad_norm(self) -> float:
        """Global gradient norm of the last step (host read: a sync)."""
        return float(self.gnorm_sq.sqrt().item())

    def step(self, input_ids, labels, attention_mask=None, image_features=None, slot_rows=None):
        loss, count = self.forward_backward(input_ids, labels, attention_mask, image_features, slot_rows)
        self.optimizer_step()
        return loss, count

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {n: t.detach().clone() for n, t in self.p.items()}
