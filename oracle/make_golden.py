"""Generate tests/golden/*.npz by running the REFERENCE's own code (through oracle/ref_shim.py) on seeded synthetic
weights/inputs.  Run in the build container only:   python -m oracle.make_golden

The goldens pin (a) the oracle restatement (tests/test_oracle.py, CPU) and (b) the HIP path (tests/test_model_gpu.py):
  - image features                     LlavaLlamaForCausalLM.encode_images          llava/model/llava_arch.py:94-97
  - splice: embeds/mask/pos/labels     prepare_inputs_labels_for_multimodal         llava/model/llava_arch.py:99-240
  - logits (all positions)             LlavaLlamaForCausalLM.forward                llava/model/language_model/llava_llama.py:56-99
  - greedy token ids with KV cache     generate(do_sample=False, use_cache=True)    llava/serve/model_worker.py:174-185
  - tokenizer_image_token KATs         llava/mm_utils.py:47-67
Weights and inputs are NOT stored: they are regenerated from synthetic/recipes.py (config name, seed).
"""
from __future__ import annotations

import json
import os
from dataclasses import replace

import numpy as np
import torch

from synthetic import recipes as synth

from . import ref_shim

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SEED = 0


def cases_for(cfg):
    """name -> dict(input_ids [B,L], attention_mask|None, labels|None, n_images, images_as_list, cfg overrides, pass_pos)."""
    V = cfg.vocab_size
    rs = np.random.RandomState(7)

    def row(n, img_at=()):
        r = rs.randint(3, V, size=(n,)).astype(np.int64)
        r[0] = 1
        for p in img_at:
            r[p] = synth.IMAGE_TOKEN_INDEX
        return r

    c = {}
    c["single"] = dict(ids=row(12, (5,))[None], mask=None, labels=None, n_images=1)
    # SURVEY Appendix B2 shape: row 0 = image + right padding, row 1 = text-only (still consumes image slot 1)
    ids = np.stack([np.concatenate([row(5, (2,)), np.zeros(2, np.int64)]), row(7)])
    mask = np.array([[1, 1, 1, 1, 1, 0, 0], [1] * 7], np.int64)
    c["batch_mixed"] = dict(ids=ids, mask=mask, labels=ids.copy(), n_images=2, pass_pos=True)
    c["batch_left_pad"] = dict(ids=ids, mask=mask, labels=ids.copy(), n_images=2, cfg=dict(tokenizer_padding_side="left"))
    c["truncate"] = dict(ids=row(14, (3,))[None], mask=np.ones((1, 14), np.int64), labels=None, n_images=1,
                         cfg=dict(tokenizer_model_max_length=cfg.tokens_per_image + 5))
    c["two_images"] = dict(ids=row(9, (2, 6))[None], mask=None, labels=row(9)[None], n_images=2)
    c["images_list"] = dict(ids=np.stack([row(8, (1,)), row(8, (4,))]), mask=np.ones((2, 8), np.int64), labels=None, n_images=3,
                            images_as_list=[2, 1])    # entry 0 has 2 crops -> 2P rows in one slot (llava_arch.py:114-119)
    return c


def run_config(name: str) -> None:
    base = synth.CONFIGS[name]
    weights = synth.make_weights(base, SEED)
    out = {}
    meta = {"config": name, "seed": SEED, "transformers": None, "cases": {}}
    for cname, case in cases_for(base).items():
        cfg = replace(base, **case.get("cfg", {}))
        model = ref_shim.build_reference_model(cfg, weights)
        meta["transformers"] = ref_shim.load_reference().transformers_version
        ids = torch.from_numpy(case["ids"])
        mask = None if case["mask"] is None else torch.from_numpy(case["mask"])
        labels = None if case["labels"] is None else torch.from_numpy(case["labels"])
        pix = torch.from_numpy(synth.make_pixels(cfg, case["n_images"], seed=1))
        if case.get("images_as_list"):
            chunks, o = [], 0
            for n in case["images_as_list"]:
                chunks.append(pix[o:o + n]); o += n
            images = chunks
        else:
            images = pix
        pos_in = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1) if case.get("pass_pos") else None
        with torch.no_grad():
            feats = model.encode_images(pix)
            r = model.prepare_inputs_labels_for_multimodal(ids, pos_in, mask, None, labels, images)
            _, pos, am, _, embeds, new_labels = r
            fw = model(input_ids=ids, attention_mask=mask, labels=None, images=images, use_cache=True)
        p = f"{cname}."
        out[p + "input_ids"] = case["ids"]
        if case["mask"] is not None:
            out[p + "attention_mask_in"] = case["mask"]
        if case["labels"] is not None:
            out[p + "labels_in"] = case["labels"]
        out[p + "image_features"] = feats.numpy()
        out[p + "inputs_embeds"] = embeds.numpy()
        out[p + "logits"] = fw.logits.float().numpy()
        if am is not None:
            out[p + "attention_mask"] = am.numpy()
        if pos is not None:
            out[p + "position_ids"] = pos.numpy()
        if new_labels is not None:
            out[p + "labels"] = new_labels.numpy()
        meta["cases"][cname] = dict(n_images=case["n_images"], images_as_list=case.get("images_as_list"), cfg=case.get("cfg", {}),
                                    pass_pos=bool(case.get("pass_pos")), returned_none=dict(mask=am is None, pos=pos is None, labels=new_labels is None))
        if cname == "single":
            with torch.no_grad():
                gen = model.generate(inputs=ids, images=images, do_sample=False, max_new_tokens=12, use_cache=True,
                                     past_key_values=ref_shim.subscriptable_cache())
            out[p + "generate"] = gen.numpy()            # echoes the input ids (incl. -200) then the new tokens
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(OUT_DIR, f"{name}.npz"), **out)
    print(f"wrote {name}.npz: {len(out)} arrays, {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB raw")


def hidden_states_golden() -> None:
    """`output_hidden_states=True` of LlavaLlamaForCausalLM.forward (llava_llama.py:63-64, 88-99 -> LlamaModel's all_hidden_states): the tuple of
    L + 1 tensors for two cases of the tiny configs -> tests/golden/hidden_states.npz (own file: the other goldens stay byte-identical)."""
    out = {}
    for name in ("tiny", "tiny_gqa"):
        base = synth.CONFIGS[name]
        weights = synth.make_weights(base, SEED)
        cases = cases_for(base)
        for cname in ("single", "batch_mixed"):
            case = cases[cname]
            model = ref_shim.build_reference_model(base, weights)
            ids = torch.from_numpy(case["ids"])
            mask = None if case["mask"] is None else torch.from_numpy(case["mask"])
            pix = torch.from_numpy(synth.make_pixels(base, case["n_images"], seed=1))
            with torch.no_grad():
                fw = model(input_ids=ids, attention_mask=mask, images=pix, use_cache=True, output_hidden_states=True)
                r = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, pix)
            assert len(fw.hidden_states) == base.num_hidden_layers + 1
            out[f"{name}.{cname}.hidden_states"] = torch.stack([h.float() for h in fw.hidden_states]).numpy()       # [L + 1, B, T, H]
            out[f"{name}.{cname}.logits"] = fw.logits.float().numpy()
            if r[2] is not None:
                out[f"{name}.{cname}.attention_mask"] = r[2].numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "hidden_states.npz"), **out)
    print(f"wrote hidden_states.npz: {len(out)} arrays, {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB raw")


def attentions_golden() -> None:
    """`output_attentions=True` of LlavaLlamaForCausalLM.forward (llava_llama.py:62-63, 88-99 -> LlamaModel's all_self_attns, eager attention): the tuple of
    L tensors [B, heads, T, T] for two cases of the tiny configs, and one cached decode step behind the first case ([B, heads, 1, T + 1])
    -> tests/golden/attentions.npz (own file: the other goldens stay byte-identical)."""
    out = {}
    for name in ("tiny", "tiny_gqa"):
        base = synth.CONFIGS[name]
        weights = synth.make_weights(base, SEED)
        cases = cases_for(base)
        for cname in ("single", "batch_mixed"):
            case = cases[cname]
            model = ref_shim.build_reference_model(base, weights)
            ids = torch.from_numpy(case["ids"])
            mask = None if case["mask"] is None else torch.from_numpy(case["mask"])
            pix = torch.from_numpy(synth.make_pixels(base, case["n_images"], seed=1))
            with torch.no_grad():
                fw = model(input_ids=ids, attention_mask=mask, images=pix, use_cache=True, output_attentions=True)
                r = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, None, pix)
            assert len(fw.attentions) == base.num_hidden_layers
            out[f"{name}.{cname}.attentions"] = torch.stack([a.float() for a in fw.attentions]).numpy()             # [L, B, heads, T, T]
            out[f"{name}.{cname}.logits"] = fw.logits.float().numpy()
            if r[2] is not None:
                out[f"{name}.{cname}.attention_mask"] = r[2].numpy()
            if cname == "single":
                nxt = fw.logits[:, -1].argmax(-1, keepdim=True)
                with torch.no_grad():
                    st = model(input_ids=nxt, past_key_values=fw.past_key_values, use_cache=True, output_attentions=True)
                out[f"{name}.{cname}.next_id"] = nxt.numpy()
                out[f"{name}.{cname}.step_attentions"] = torch.stack([a.float() for a in st.attentions]).numpy()    # [L, B, heads, 1, T + 1]
    np.savez_compressed(os.path.join(OUT_DIR, "attentions.npz"), **out)
    print(f"wrote attentions.npz: {len(out)} arrays, {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB raw")


def tokenizer_kats() -> None:
    """llava/mm_utils.py:47-67 with the fake tokenizer of SURVEY Appendix B1."""
    ref = ref_shim.load_reference()

    class FakeTok:
        bos_token_id = 1

        def __call__(self, text):
            class R:
                pass
            r = R()
            r.input_ids = [1] + [10 + ord(c) % 50 for c in text]
            return r

    prompts = ["AB<image>\nCD", "<image>\nX", "no image", "A<image>B<image>C"]
    res = {p: ref.mm_utils.tokenizer_image_token(p, FakeTok()) for p in prompts}
    with open(os.path.join(OUT_DIR, "tokenizer_image_token.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote tokenizer_image_token.json", res)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for n in ("tiny", "tiny_gqa"):
        run_config(n)
    hidden_states_golden()
    attentions_golden()
    tokenizer_kats()
