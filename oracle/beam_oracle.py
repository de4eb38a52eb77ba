"""TEST INFRASTRUCTURE — CPU restatement of beam search (`generate(num_beams > 1)`) over the oracle's forward pass.

Follows GenerationMixin.beam_search + BeamSearchScorer.process / finalize + BeamHypotheses of transformers 4.31 (the release the reference pins,
pyproject.toml:16; un-vendored third-party code, restated from its published algorithm): log_softmax + running beam scores, top 2 * num_beams over
the flattened [num_beams * V] scores, EOS candidates among the first num_beams close hypotheses scored sum_logprobs / len(hypothesis) ** length_penalty
(len counts the prompt in 4.31), the rest continue; early_stopping=False stops when the worst kept hypothesis beats best_sum_logprobs / cur_len **
length_penalty.  No KV cache: every step re-runs the whole forward per beam (tiny configs only).
Pinned: tests/test_beam_oracle_vs_reference.py runs it against the live reference model's own generate(num_beams=...) (no-EOS cases, where every
release agrees); EOS-terminated hypotheses follow the 4.31 rules above and are not pinned against the installed (newer) transformers, whose length
normalisation differs.  `stopping_criteria` likewise follow 4.31 (`if beam_scorer.is_done or stopping_criteria(input_ids, scores): break`, the list being
an `any` over its members, evaluated on the [num_beams, cur_len] ids of the continuing beams) — PARITY UNPINNED for that argument: the installed release
applies criteria per candidate instead."""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import llava_oracle as O


def beam_search(w, cfg, input_ids: torch.Tensor, images, num_beams: int, max_new_tokens: int, eos_ids: Sequence[int] = (), length_penalty: float = 1.0,
                early_stopping=False, length_counts_prompt: bool = True, stopping_criteria=None) -> List[int]:
    B, V = int(num_beams), cfg.vocab_size
    eos = set(int(e) for e in eos_ids)
    prompt = input_ids[0].tolist()
    beams = [[] for _ in range(B)]
    scores = torch.full((B,), -1e9); scores[0] = 0.0
    hyps: List[tuple] = []                      # (score, tokens)
    base = len(prompt) if length_counts_prompt else 0

    def add(tokens, sum_lp, length):
        s = sum_lp / (max(length, 1) ** length_penalty)
        if len(hyps) < B or s > min(h[0] for h in hyps):
            hyps.append((s, list(tokens)))
            if len(hyps) > B:
                hyps.remove(min(hyps, key=lambda h: h[0]))

    done = False
    for t in range(max_new_tokens):
        rows = []
        for b in range(B):
            ids_b = torch.tensor([prompt + beams[b]], dtype=torch.long)
            with torch.no_grad():
                logits = O.llava_forward(w, cfg, ids_b, images, last_only=True)[0][0, -1].float()
            rows.append(torch.log_softmax(logits, dim=-1) + scores[b])
        flat = torch.cat(rows)
        top, idx = torch.topk(flat, 2 * B, largest=True, sorted=True)
        cur_len = base + t
        nxt = []
        for rank in range(2 * B):
            b, tok, sc = int(idx[rank]) // V, int(idx[rank]) % V, float(top[rank])
            if tok in eos:
                if rank >= B:
                    continue
                add(beams[b], sc, cur_len)
            else:
                nxt.append((sc, tok, b))
            if len(nxt) == B:
                break
        assert len(nxt) == B
        if len(hyps) >= B and not done:
            if early_stopping is True:
                done = True
            else:
                done = min(h[0] for h in hyps) >= float(top[0]) / (max(cur_len, 1) ** length_penalty)
        scores = torch.tensor([n[0] for n in nxt])
        beams = [beams[b] + [tok] for _, tok, b in nxt]
        if done:
            break
        if stopping_criteria:
            full = torch.tensor([prompt + bm for bm in beams], dtype=torch.long)
            if any(bool(c(full, None)) for c in stopping_criteria):
                break
    if not done:
        for b in range(B):
            add(beams[b], float(scores[b]), base + len(beams[b]))
    out = list(max(hyps, key=lambda h: h[0])[1])
    if len(out) < max_new_tokens and eos:
        out.append(int(list(eos_ids)[0]))
    return out
