"""Beam search (`generate(num_beams > 1)`), passed through by the reference's eval scripts (llava/eval/run_llava.py:121, model_vqa_loader.py:104).

The algorithm is GenerationMixin.beam_search + BeamSearchScorer of the reference's transformers (4.31): per step
    scores = log_softmax(logits) + beam_scores[:, None];  the 2 * num_beams best (score, token, source beam) over all beams;
    walk them best-first: an EOS candidate among the first num_beams closes a hypothesis (score = sum_logprobs / length ** length_penalty), the
    others fill the next num_beams beams;  stop when the kept hypotheses cannot be beaten (early_stopping rules) or at max length;
    finalize: open beams become hypotheses too, the best one is returned (with EOS appended when it ended early).
Here the device does the heavy half — one decode step for all beams together (lmx_decode_batch), log-softmax + top-K per beam row
(lmx_op_beam_topk) — and the KV caches follow the beams: a surviving beam keeps its sequence, only duplicated beams are copied (lmx_seq_copy).
The hypothesis bookkeeping is a few integers per step and stays on the host, as in the reference.

Beam SAMPLE (`num_beams > 1` with `do_sample=True`: GenerationMixin.beam_sample, what run_llava.py:115-125 asks for when --num_beams is raised at its default
temperature) is the same loop with a different candidate producer: scores = warpers(log_softmax + beam_scores) (temperature divides the summed score, top-k /
top-p keep a survivor set per beam row with min_tokens_to_keep = 2), then 2 * num_beams draws WITHOUT replacement from softmax over the num_beams x V block,
ranked by score.  The device draws them as the top-2B of score + Gumbel noise (lmx_op_beam_sample_topk: Plackett-Luce order, the distribution of
torch.multinomial(replacement=False)); ids for a given seed are not comparable with a torch run (different generator), the distribution is.

`length_counts_prompt`: transformers 4.31 divides by the FULL length of the hypothesis (prompt included: BeamHypotheses.add uses hyp.shape[-1]);
later releases use the generated length only.  The default follows the reference's pinned release."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Set, Tuple

import torch

from ._C import check, lib, ptr, stream_handle, torch_dtype_code


class _Hypotheses:
    """BeamHypotheses: the num_beams best finished hypotheses of one batch row."""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, float(length_penalty), early_stopping
        self.beams: List[Tuple[float, List[int]]] = []
        self.worst = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, tokens: List[int], sum_logprobs: float, length: int) -> None:
        score = sum_logprobs / (max(length, 1) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst:
            self.beams.append((score, list(tokens)))
            if len(self) > self.num_beams:
                self.beams.remove(min(self.beams, key=lambda b: b[0]))
            self.worst = min(b[0] for b in self.beams)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        if self.early_stopping is False:
            return self.worst >= best_sum_logprobs / (max(cur_len, 1) ** self.length_penalty)
        raise NotImplementedError('early_stopping="never" is not supported')


def rank_draws(keys: torch.Tensor, scores: torch.Tensor, ids: torch.Tensor, V: int, K: int) -> List[Tuple[float, int, int]]:
    """Host half of a beam-sample step.  keys / scores / ids [num_beams, K]: per beam row the K largest (score + Gumbel) keys with their scores and token
    ids (lmx_op_beam_sample_topk; id -1 = the row had fewer survivors).  The K largest keys over all rows ARE the 2 * num_beams draws without replacement
    from softmax over the num_beams x V block; GenerationMixin.beam_sample then ranks the drawn candidates by score (torch.sort(descending)).
    Returns [(score, token, beam)] in that order (ties: lower flat index beam * V + token first, the order a stable sort of the flat block gives)."""
    B = keys.shape[0]
    drawn = sorted(((float(keys[b, k]), float(scores[b, k]), int(ids[b, k]), b) for b in range(B) for k in range(keys.shape[1]) if int(ids[b, k]) >= 0),
                   key=lambda c: (-c[0], c[3] * V + c[2]))[:K]
    return sorted(((c[1], c[2], c[3]) for c in drawn), key=lambda c: (-c[0], c[2] * V + c[1]))


def beam_search(model, ids: torch.Tensor, images, attention_mask, num_beams: int, max_new_tokens: int, eos_set: Set[int], length_penalty: float = 1.0,
                early_stopping=False, prefill_chunk: int = 0, length_counts_prompt: bool = True, eos_first: Optional[int] = None,
                stopping_criteria=None, sample: Optional[dict] = None) -> List[int]:
    """ids [1, L] (with image markers).  Returns the generated ids of the best hypothesis (EOS included when it ended early).
    stopping_criteria: evaluated as GenerationMixin.beam_search does (transformers 4.31 generation/utils.py: `if beam_scorer.is_done or
    stopping_criteria(input_ids, scores): break`) — after every step, on the [num_beams, L + t] ids of the beams that continue; the loop then ends
    and finalize() ranks the open beams with the finished hypotheses.
    sample: None = beam search; dict(temperature, top_p, top_k, seed) = beam sample (see the module docstring)."""
    from .batching import DecodeBatch
    from .model import LmxKVCache
    B, dev, dt = int(num_beams), model.device, model.dtype
    V, Vpitch = model.config.vocab_size, model._vocab_cap
    K = 2 * B
    cache0, logits0 = model._prefill_request(ids, images, attention_mask, None, prefill_chunk, return_logits=True)
    caches = [cache0] + [LmxKVCache(model, 1) for _ in range(B - 1)]
    batch = DecodeBatch(model, B)
    try:
        prompt_len = lib.lmx_seq_length(cache0.seqs[0])           # positions after the image splice (what the reference's input_ids length would be is ids.shape[1])
        base_len = int(ids.shape[1]) if length_counts_prompt else 0
        for c in caches[1:]:
            check(lib.lmx_seq_copy(c.seqs[0], cache0.seqs[0], stream_handle()), "lmx_seq_copy")
        room = model.s_max - prompt_len
        steps = min(int(max_new_tokens), room)
        hyps = _Hypotheses(B, length_penalty, early_stopping)
        beam_scores = torch.full((B,), -1e9, dtype=torch.float32); beam_scores[0] = 0.0
        tokens: List[List[int]] = [[] for _ in range(B)]
        logits = logits0.reshape(1, Vpitch).expand(B, Vpitch).contiguous()              # every beam starts from the prompt's last position
        sc = torch.empty((B, K), dtype=torch.float32, device=dev); ix = torch.empty((B, K), dtype=torch.int32, device=dev)
        if sample is not None:
            s_temp, s_seed = float(sample["temperature"]), int(sample["seed"])
            s_top_p = float(sample["top_p"]) if sample.get("top_p") is not None else 1.0
            s_top_k = int(sample.get("top_k") or 0)
            warp = s_top_k > 0 or s_top_p < 1.0
            keys = torch.empty((B, K), dtype=torch.float32, device=dev)
            keep = torch.empty((B, V), dtype=torch.uint8, device=dev) if warp else None
            sc2 = torch.empty((B, 2), dtype=torch.float32, device=dev); ix2 = torch.empty((B, 2), dtype=torch.int32, device=dev)
            scratch_tok = torch.empty((1,), dtype=torch.long, device=dev)
        done = False
        for t in range(steps):
            bs_dev = beam_scores.to(dev)
            if sample is None:
                check(lib.lmx_op_beam_topk(torch_dtype_code(dt), ptr(logits), logits.stride(0), V, B, ptr(bs_dev), K, ptr(sc), ptr(ix), stream_handle()), "lmx_op_beam_topk")
                sc_h, ix_h = sc.cpu(), ix.cpu()
                cands = sorted(((float(sc_h[b, k]), int(ix_h[b, k]), b) for b in range(B) for k in range(K) if int(ix_h[b, k]) >= 0),
                               key=lambda c: (-c[0], c[2] * V + c[1]))[:K]
            else:
                if warp:
                    # survivor set of TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper per beam row (a per-row constant — the beam score — does not
                    # change it); with num_beams > 1 transformers builds both with min_tokens_to_keep = 2 (generation/utils.py: _get_logits_warper)
                    for b in range(B):
                        check(lib.lmx_op_sample(torch_dtype_code(dt), ptr(logits[b]), V, s_temp, s_top_p, max(s_top_k, 2) if s_top_k > 0 else 0, s_seed, None, None,
                                                ptr(scratch_tok), ptr(keep[b]), stream_handle()), "lmx_op_sample")
                    check(lib.lmx_op_beam_topk(torch_dtype_code(dt), ptr(logits), logits.stride(0), V, B, None, 2, ptr(sc2), ptr(ix2), stream_handle()), "lmx_op_beam_topk")
                    keep.scatter_(1, ix2.long(), 1)
                check(lib.lmx_op_beam_sample_topk(torch_dtype_code(dt), ptr(logits), logits.stride(0), V, B, ptr(keep) if warp else None, ptr(bs_dev), s_temp, s_seed,
                                                  (t * B * V) & 0xFFFFFFFF, K, ptr(keys), ptr(sc), ptr(ix), stream_handle()), "lmx_op_beam_sample_topk")
                cands = rank_draws(keys.cpu(), sc.cpu(), ix.cpu(), V, K)
            cur_len = base_len + t                                   # length of a hypothesis before this step's token (transformers 4.31 `cur_len`)
            nxt: List[Tuple[float, int, int]] = []
            for rank, (score, tok, b) in enumerate(cands):
                if tok in eos_set:
                    if rank >= B:
                        continue
                    hyps.add(tokens[b], score, cur_len)
                else:
                    nxt.append((score, tok, b))
                if len(nxt) == B:
                    break
            if len(nxt) < B:
                raise ValueError("beam search: fewer than num_beams non-EOS candidates (the reference raises here as well)")
            done = done or hyps.is_done(cands[0][0], cur_len)
            # ---- the next beams: scores, tokens, caches (a source beam used once keeps its sequence; further uses get copies) ----------------
            beam_scores = torch.tensor([n[0] for n in nxt], dtype=torch.float32)
            src = [n[2] for n in nxt]
            new_tokens = [tokens[b] + [tok] for _, tok, b in nxt]
            new_caches: List[Optional[LmxKVCache]] = [None] * B
            used = set()
            for j, b in enumerate(src):
                if b not in used:
                    new_caches[j] = caches[b]; used.add(b)
            free = [caches[b] for b in range(B) if b not in used]
            for j, b in enumerate(src):
                if new_caches[j] is None:
                    c = free.pop()
                    check(lib.lmx_seq_copy(c.seqs[0], caches[b].seqs[0], stream_handle()), "lmx_seq_copy")
                    new_caches[j] = c
            caches, tokens = new_caches, new_tokens
            if not done and stopping_criteria:
                full = torch.cat([ids.cpu().expand(B, -1), torch.tensor(tokens, dtype=torch.long)], dim=1)
                if any(bool(c(full, None)) for c in stopping_criteria):
                    break                                            # not `done`: the open beams are ranked in finalize
            if done or t + 1 == steps:
                break
            logits = torch.empty((B, Vpitch), dtype=dt, device=dev)
            batch.step([c.seqs[0] for c in caches], tokens=[tk[-1] for tk in tokens], n_steps=1, greedy=True, logits=logits, want_ids=False)
        # ---- finalize: open beams are hypotheses too; best score wins --------------------------------------------------------------------
        if not done:
            for b in range(B):
                hyps.add(tokens[b], float(beam_scores[b]), base_len + len(tokens[b]))
        out = list(max(hyps.beams, key=lambda h: h[0])[1])
        real_eos = sorted(e for e in eos_set if e >= 0)       # eos_token_id = -1 is this API's "no EOS": nothing to append then (transformers: eos_token_id None)
        if len(out) < steps and real_eos:            # finalize(): a hypothesis shorter than max_length gets eos_token_id[0] appended
            out.append(int(eos_first) if eos_first is not None else real_eos[0])
        return out
    finally:
        batch.close()
        for c in caches:
            if c is not None:
                c.close()
