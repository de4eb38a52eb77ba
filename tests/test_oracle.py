"""CPU tests: the oracle restatement and the host half of the C ABI against golden vectors produced by the
reference's own code (oracle/make_golden.py).  Integer outputs are bit-exact; fp32 tensors within 2e-5."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, case_inputs, load, split_images

from oracle import llava_oracle as O
from synthetic import recipes as synth

CONFIGS = ["tiny", "tiny_gqa"]
CASES = ["single", "batch_mixed", "batch_left_pad", "truncate", "two_images", "images_list"]
FP32_TOL = 2e-5


@pytest.fixture(scope="module", params=CONFIGS)
def golden(request):
    z, meta = load(request.param)
    w = O.to_torch_weights(synth.make_weights(synth.CONFIGS[request.param], meta["seed"]))
    return z, meta, w


@pytest.mark.parametrize("cname", CASES)
def test_oracle_matches_reference_golden(golden, cname):
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, cname)
    p = cname + "."
    pix_t = torch.from_numpy(pix)
    feats = O.encode_images(w, cfg, pix_t)
    assert np.abs(feats.numpy() - z[p + "image_features"]).max() < FP32_TOL
    images = split_images(pix_t, cm["images_as_list"])
    ids_t = torch.from_numpy(ids)
    mask_t = None if mask is None else torch.from_numpy(mask)
    lab_t = None if labels is None else torch.from_numpy(labels)
    pos_in = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1) if cm["pass_pos"] else None
    _, pos, am, _, emb, new_lab = O.prepare_inputs_labels_for_multimodal(w, cfg, ids_t, pos_in, mask_t, None, lab_t, images)
    assert emb.shape == z[p + "inputs_embeds"].shape
    assert np.abs(emb.numpy() - z[p + "inputs_embeds"]).max() < FP32_TOL
    # None-passthrough (llava_arch.py:227-238) and bit-exact integers
    assert (am is None) == cm["returned_none"]["mask"] and (pos is None) == cm["returned_none"]["pos"] and (new_lab is None) == cm["returned_none"]["labels"]
    if am is not None:
        assert np.array_equal(am.numpy(), z[p + "attention_mask"])
    if pos is not None:
        assert np.array_equal(pos.numpy(), z[p + "position_ids"])
    if new_lab is not None:
        assert np.array_equal(new_lab.numpy(), z[p + "labels"])
    logits, _, _, _ = O.llava_forward(w, cfg, ids_t, images, attention_mask=mask_t)
    ref = z[p + "logits"]
    valid = np.ones(ref.shape[:2], bool) if am is None else am.numpy().astype(bool)
    assert np.abs(logits.numpy() - ref)[valid].max() < FP32_TOL * 5


@pytest.mark.parametrize("cname", ["single", "batch_mixed"])
def test_oracle_hidden_states_match_reference_golden(golden, cname):
    """`output_hidden_states=True` (llava_llama.py:63-64 passes it to LlamaModel): the reference's tuple of L + 1 tensors (tests/golden/hidden_states.npz,
    oracle/make_golden.py::hidden_states_golden) against the oracle's, on the rows the attention mask keeps."""
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, cname)
    hz = np.load(os.path.join(GOLDEN_DIR, "hidden_states.npz"))
    key = f"{meta['config']}.{cname}."
    ref = hz[key + "hidden_states"]                                  # [L + 1, B, T, H]
    assert np.array_equal(hz[key + "logits"], z[cname + ".logits"])  # the same forward as the main golden
    hs = []
    O.llava_forward(w, cfg, torch.from_numpy(ids), torch.from_numpy(pix), attention_mask=None if mask is None else torch.from_numpy(mask), hidden_out=hs)
    assert len(hs) == cfg.num_hidden_layers + 1 == ref.shape[0]
    valid = np.ones(ref.shape[1:3], bool) if key + "attention_mask" not in hz.files else hz[key + "attention_mask"].astype(bool)
    for l, h in enumerate(hs):
        assert h.shape == ref[l].shape
        assert np.abs(h.numpy() - ref[l])[valid].max() < FP32_TOL * 5, l


@pytest.mark.parametrize("cname", ["single", "batch_mixed"])
def test_oracle_attentions_match_reference_golden(golden, cname):
    """`output_attentions=True` (llava_llama.py:62-63 passes it to LlamaModel; eager attention returns the post-softmax weights): the reference's tuple of L
    tensors [B, heads, T, T] (tests/golden/attentions.npz, oracle/make_golden.py::attentions_golden) against the oracle's, on the query rows and key columns the
    attention mask keeps; and one cached decode step behind the un-padded case ([B, heads, 1, T + 1])."""
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, cname)
    az = np.load(os.path.join(GOLDEN_DIR, "attentions.npz"))
    key = f"{meta['config']}.{cname}."
    ref = az[key + "attentions"]                                     # [L, B, heads, T, T]
    assert np.array_equal(az[key + "logits"], z[cname + ".logits"])  # the same forward as the main golden
    att = []
    _, past, _, _ = O.llava_forward(w, cfg, torch.from_numpy(ids), torch.from_numpy(pix), attention_mask=None if mask is None else torch.from_numpy(mask), attn_out=att)
    assert len(att) == cfg.num_hidden_layers == ref.shape[0]
    valid = np.ones(ref.shape[1:2] + ref.shape[3:4], bool) if key + "attention_mask" not in az.files else az[key + "attention_mask"].astype(bool)
    pair = valid[:, None, :, None] & valid[:, None, None, :]         # [B, 1, T, T]: kept query rows x kept key columns
    for l, a in enumerate(att):
        assert a.shape == ref[l].shape
        d = np.abs(a.numpy() - ref[l])
        assert d[np.broadcast_to(pair, d.shape)].max() < FP32_TOL, l
        rows = a.numpy().sum(-1)
        assert np.abs(rows - 1.0)[np.broadcast_to(valid[:, None, :], rows.shape)].max() < 1e-5
    if cname == "single":
        step = []
        nxt = torch.from_numpy(az[key + "next_id"])
        O.llama_forward(w, cfg, w["model.embed_tokens.weight"][nxt], past=past, attn_out=step)
        sref = az[key + "step_attentions"]
        assert len(step) == sref.shape[0]
        for l, a in enumerate(step):
            assert a.shape == sref[l].shape and np.abs(a.numpy() - sref[l]).max() < FP32_TOL, l


def test_oracle_greedy_matches_reference_generate(golden):
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, "single")
    gen = z["single.generate"]
    assert np.array_equal(gen[0, : ids.shape[1]], ids[0])          # generate echoes the prompt incl. -200
    new = gen[0, ids.shape[1]:].tolist()
    got = O.greedy_generate(w, cfg, torch.from_numpy(ids), torch.from_numpy(pix), len(new))
    assert got == new


def test_oracle_cache_equals_full_recompute(golden):
    """size-independent property: greedy with the KV cache == re-running the whole prefix each step."""
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, "single")
    ids_t, pix_t = torch.from_numpy(ids), torch.from_numpy(pix)
    cached = O.greedy_generate(w, cfg, ids_t, pix_t, 5)
    _, _, emb, _ = O.llava_forward(w, cfg, ids_t, pix_t)
    toks = []
    for _ in range(5):
        logits, _ = O.llama_forward(w, cfg, emb, last_only=True)
        t = int(torch.argmax(logits[0, -1]))
        toks.append(t)
        emb = torch.cat([emb, w["model.embed_tokens.weight"][torch.tensor([[t]])]], dim=1)
    assert toks == cached


def test_tokenizer_image_token_kats():
    """llava/mm_utils.py:47-67 KATs against the host-side mirror in the product package."""
    from llava_mi355x.mm_utils import tokenizer_image_token

    class FakeTok:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass
            r = R(); r.input_ids = [1] + [10 + ord(c) % 50 for c in text]
            return r

    kats = json.load(open(os.path.join(GOLDEN_DIR, "tokenizer_image_token.json")))
    for prompt, want in kats.items():
        assert tokenizer_image_token(prompt, FakeTok()) == want


# ---- host half of the C ABI (no GPU needed) -------------------------------------------------------------------------
def _lib():
    from llava_mi355x import _C
    return _C


def test_cabi_loads_and_exports_every_declared_symbol():
    _C = _lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "llava_mi355x.h")).read()
    declared = sorted(set(re.findall(r"\b(lmx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(_C.lib, name), f"{name} declared in include/llava_mi355x.h but not exported"
        assert name in _C.EXPORTED, f"{name} has no ctypes signature"
    assert _C.lib.lmx_abi_version() == _C.LMX_ABI_VERSION


def _plan(ids, mask, labels, P, slot_rows, n_slots, max_len, left):
    _C = _lib()
    ids = np.ascontiguousarray(ids, np.int64)
    B, L = ids.shape
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    lb = None if labels is None else np.ascontiguousarray(labels, np.int64)
    sr = None if slot_rows is None else np.ascontiguousarray(slot_rows, np.int32)
    vp = lambda a: ctypes.c_void_p(0) if a is None else ctypes.c_void_p(a.ctypes.data)
    T = ctypes.c_int32(0)
    rc = _C.lib.lmx_splice_plan(vp(ids), vp(m), vp(lb), B, L, P, vp(sr), n_slots, max_len, int(left), ctypes.byref(T), None, None, None, None)
    if rc:
        raise _C.LmxError(_C.last_error())
    src = np.zeros((B, T.value), np.int32); om = np.zeros((B, T.value), np.uint8)
    op = np.zeros((B, T.value), np.int64); ol = np.zeros((B, T.value), np.int64)
    _C.check(_C.lib.lmx_splice_plan(vp(ids), vp(m), vp(lb), B, L, P, vp(sr), n_slots, max_len, int(left), ctypes.byref(T),
                                    vp(src), vp(om), vp(op), vp(ol)))
    return src, om, op, ol


@pytest.mark.parametrize("cname", CASES)
def test_cabi_splice_plan_bit_exact(golden, cname):
    z, meta, w = golden
    cfg, cm, ids, mask, labels, pix = case_inputs(z, meta, cname)
    p = cname + "."
    P = cfg.tokens_per_image
    slot_rows = [n * P for n in cm["images_as_list"]] if cm["images_as_list"] else None
    n_slots = len(slot_rows) if slot_rows else cm["n_images"]
    src, om, op, ol = _plan(ids, mask, labels, P, slot_rows, n_slots, cfg.tokenizer_model_max_length or 0, cfg.tokenizer_padding_side == "left")
    # the same plan from the numpy restatement
    s2, m2, p2, l2 = O.splice_plan(ids, None if mask is None else mask.astype(bool), labels, slot_rows or [P] * n_slots,
                                   cfg.tokenizer_model_max_length, cfg.tokenizer_padding_side == "left")
    assert np.array_equal(src, s2) and np.array_equal(om.astype(bool), m2) and np.array_equal(op, p2) and np.array_equal(ol, l2)
    # and against the reference's own outputs
    assert src.shape == z[p + "inputs_embeds"].shape[:2]
    if p + "attention_mask" in z.files:
        assert np.array_equal(om, z[p + "attention_mask"].astype(np.uint8))
    if p + "position_ids" in z.files:
        assert np.array_equal(op, z[p + "position_ids"])
    if p + "labels" in z.files:
        assert np.array_equal(ol, z[p + "labels"])
    # image-token indices bit-exact: rows the reference filled with image features are exactly the src <= -2 rows
    emb = z[p + "inputs_embeds"]; feats = z[p + "image_features"].reshape(-1, emb.shape[-1])
    for b in range(src.shape[0]):
        for t in range(src.shape[1]):
            if src[b, t] <= -2:
                assert np.array_equal(emb[b, t], feats[-2 - src[b, t]])
            elif src[b, t] == -1:
                assert not emb[b, t].any()


def test_cabi_splice_plan_edge_cases():
    _C = _lib()
    # SURVEY Appendix B2 known answer (4 patches/image)
    ids = np.array([[1, 5, -200, 7, 8, 0, 0], [1, 5, 6, 7, 8, 9, 4]]); mask = np.array([[1, 1, 1, 1, 1, 0, 0], [1] * 7])
    src, om, op, ol = _plan(ids, mask, ids.copy(), 4, None, 2, 0, False)
    assert src.tolist() == [[1, 5, -2, -3, -4, -5, 7, 8], [1, 5, 6, 7, 8, 9, 4, -1]]
    assert om.tolist() == [[1] * 8, [1] * 7 + [0]]
    assert ol.tolist() == [[1, 5, -100, -100, -100, -100, 7, 8], [1, 5, 6, 7, 8, 9, 4, -100]]
    # more <image> markers than features -> error, not a crash (reference: IndexError at llava_arch.py:176)
    with pytest.raises(_C.LmxError):
        _plan(np.array([[1, -200, -200]]), None, None, 4, None, 1, 0, False)
    # text-only row with no slot left (reference: IndexError at llava_arch.py:153)
    with pytest.raises(_C.LmxError):
        _plan(np.array([[1, 2, 3]]), None, None, 4, None, 0, 0, False)
    # fully masked row -> empty sequence padded to the batch max
    src, om, op, ol = _plan(np.array([[1, 2, 3], [4, 5, 6]]), np.array([[0, 0, 0], [1, 1, 1]]), None, 4, None, 2, 0, True)
    assert src.tolist() == [[-1, -1, -1], [4, 5, 6]] and om.tolist() == [[0, 0, 0], [1, 1, 1]]
    # zero-row slot (an image entry with no crops) splices nothing
    src, _, _, _ = _plan(np.array([[1, -200, 9]]), None, None, 4, [0], 1, 0, False)
    assert src.tolist() == [[1, 9]]
