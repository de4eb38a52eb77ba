"""Randomised coverage of the integer half of prepare_inputs_labels_for_multimodal (llava/model/llava_arch.py:99-240), CPU:
  * lmx_splice_plan (host C++ behind the C ABI) vs the numpy oracle, bit-exact, on hundreds of seeded random batches
    (padding holes anywhere in the mask, several <image> markers per row, text-only rows, ragged image slots, truncation
    after expansion, left / right padding, exhausted image features -> error on both sides);
  * the oracle vs the REFERENCE's own function run on a tiny reference model (build container only: needs /root/reference):
    attention_mask / position_ids / labels bit-exact, and every output row equal to the row the plan names
    (token embedding | zero | image-feature row)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "llava-plus-codebase_amd"))

from oracle import llava_oracle as O, ref_shim  # noqa: E402
from synthetic import recipes as synth


def _random_case(rng, P, vocab=200):
    B = int(rng.randint(1, 5)); L = int(rng.randint(1, 24))
    ids = rng.randint(3, vocab, size=(B, L)).astype(np.int64)
    n_markers = 0
    for b in range(B):
        for _ in range(int(rng.randint(0, 4))):
            ids[b, rng.randint(0, L)] = -200
    mask = None
    if rng.rand() < 0.7:
        mask = (rng.rand(B, L) < 0.85).astype(np.uint8)
        style = rng.randint(0, 3)
        if style == 0:      # right padding
            for b in range(B): mask[b] = 0; mask[b, : rng.randint(0, L + 1)] = 1
        elif style == 1:    # left padding
            for b in range(B): mask[b] = 0; mask[b, L - rng.randint(0, L + 1):] = 1
    labels = ids.copy() if rng.rand() < 0.6 else None
    if labels is not None:
        labels[rng.rand(B, L) < 0.3] = -100
    keep = np.ones((B, L), bool) if mask is None else mask.astype(bool)
    need = sum(max(1, int(((ids[b] == -200) & keep[b]).sum())) for b in range(B))
    n_slots = need if rng.rand() < 0.85 else max(0, need - 1)            # sometimes one slot short -> error
    ragged = rng.rand() < 0.4
    slot_rows = [int(rng.randint(0, 3)) * P for _ in range(n_slots)] if ragged else None
    max_len = int(rng.randint(1, 40)) if rng.rand() < 0.4 else 0
    left = bool(rng.rand() < 0.5)
    return ids, mask, labels, slot_rows, n_slots, max_len, left


def _cabi_plan(_C, ids, mask, labels, P, slot_rows, n_slots, max_len, left):
    B, L = ids.shape
    T = ctypes.c_int32(0)
    vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    ids = np.ascontiguousarray(ids, np.int64)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    lb = None if labels is None else np.ascontiguousarray(labels, np.int64)
    sr = None if slot_rows is None else np.ascontiguousarray(slot_rows, np.int32)
    _C.check(_C.lib.lmx_splice_plan(vp(ids), vp(m), vp(lb), B, L, P, vp(sr), n_slots, max_len, int(left), ctypes.byref(T), None, None, None, None))
    src = np.zeros((B, T.value), np.int32); om = np.zeros((B, T.value), np.uint8)
    op = np.zeros((B, T.value), np.int64); ol = np.zeros((B, T.value), np.int64)
    _C.check(_C.lib.lmx_splice_plan(vp(ids), vp(m), vp(lb), B, L, P, vp(sr), n_slots, max_len, int(left), ctypes.byref(T),
                                    vp(src), vp(om), vp(op), vp(ol)))
    return src, om, op, ol


def test_cabi_splice_plan_equals_oracle_on_random_batches():
    from llava_mi355x import _C
    rng = np.random.RandomState(1234)
    n_err = n_ok = 0
    for _ in range(600):
        P = int(rng.randint(1, 6))
        ids, mask, labels, slot_rows, n_slots, max_len, left = _random_case(rng, P)
        rows = slot_rows if slot_rows is not None else [P] * n_slots
        try:
            want = O.splice_plan(ids, None if mask is None else mask.astype(bool), labels, rows, max_len or None, left)
        except IndexError:
            with pytest.raises(_C.LmxError):
                _cabi_plan(_C, ids, mask, labels, P, slot_rows, n_slots, max_len, left)
            n_err += 1
            continue
        src, om, op, ol = _cabi_plan(_C, ids, mask, labels, P, slot_rows, n_slots, max_len, left)
        assert np.array_equal(src, want[0]) and np.array_equal(om.astype(bool), want[1])
        assert np.array_equal(op, want[2]) and np.array_equal(ol, want[3])
        n_ok += 1
    assert n_ok > 300 and n_err > 10          # both regimes were exercised


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
@pytest.mark.parametrize("side", ["right", "left"])
def test_oracle_splice_equals_reference_on_random_batches(side):
    from dataclasses import replace
    cfg = replace(synth.CONFIGS["tiny"], tokenizer_padding_side=side, tokenizer_model_max_length=40)
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    ref_model = ref_shim.build_reference_model(cfg, wnp)
    P = cfg.tokens_per_image
    rng = np.random.RandomState(7 if side == "right" else 8)
    done = 0
    for _ in range(40):
        ids, mask, labels, _, n_slots, _, _ = _random_case(rng, P, vocab=cfg.vocab_size)
        keep = np.ones(ids.shape, bool) if mask is None else mask.astype(bool)
        need = sum(max(1, int(((ids[b] == -200) & keep[b]).sum())) for b in range(ids.shape[0]))
        pix = torch.from_numpy(synth.make_pixels(cfg, need, seed=int(rng.randint(1 << 20))))
        ids_t = torch.from_numpy(ids)
        mask_t = None if mask is None else torch.from_numpy(mask.astype(np.int64))
        lab_t = None if labels is None else torch.from_numpy(labels)
        pos_t = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1) if rng.rand() < 0.5 else None
        with torch.no_grad():
            r = ref_model.prepare_inputs_labels_for_multimodal(ids_t, pos_t, mask_t, None, lab_t, pix)
            o = O.prepare_inputs_labels_for_multimodal(w, cfg, ids_t, pos_t, mask_t, None, lab_t, pix)
        if r[4] is None:                      # L == 1: the decode-step early exit (llava_arch.py:103-112), nothing is spliced
            assert o[4] is None
            done += 1
            continue
        for a, b, what in ((r[1], o[1], "position_ids"), (r[2], o[2], "attention_mask"), (r[5], o[5], "labels")):
            assert (a is None) == (b is None), what
            if a is not None:
                assert torch.equal(a.to(torch.int64), b.to(torch.int64)), what
        assert r[4].shape == o[4].shape
        if r[4].numel():                      # every row fully masked gives an empty [B, 0, H] on both sides
            assert (r[4] - o[4]).abs().max().item() <= 2e-5
        done += 1
    assert done == 40
