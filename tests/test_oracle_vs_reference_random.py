"""CPU, build container only: the oracle (oracle/llava_oracle.py) against the REFERENCE itself on seeded random requests beyond the
committed goldens — the tiny configs (mlp2x_gelu / linear / identity / mlp3x_gelu projectors, patch / cls_patch features), random prompt lengths / image positions / batch shapes / padding: logits of
LlavaLlamaForCausalLM.forward (all positions, fp32) within 2e-5 and greedy generate() with the KV cache token for token.
This is the pinning of the oracle that tests/golden/*.npz records for six fixed cases, repeated on fresh inputs every time the
suite runs where /root/reference exists."""
import numpy as np
import pytest
import torch

from oracle import llava_oracle as O, ref_shim
from synthetic import recipes as synth

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa", "tiny_identity", "tiny_mlp3x"])
def test_forward_and_generate_match_reference(name):
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    model = ref_shim.build_reference_model(cfg, wnp)
    rng = np.random.RandomState({"tiny": 99, "tiny_gqa": 98, "tiny_identity": 97, "tiny_mlp3x": 96}[name])
    P = cfg.tokens_per_image
    for case in range(8):
        B = int(rng.randint(1, 4)); L = int(rng.randint(4, 20))
        ids = rng.randint(3, cfg.vocab_size, size=(B, L)).astype(np.int64)
        ids[:, 0] = 1
        n_img = 0
        for b in range(B):
            k = int(rng.randint(0, 3)) if B > 1 else int(rng.randint(1, 3))
            for pos in rng.choice(np.arange(1, L), size=min(k, L - 1), replace=False):
                ids[b, pos] = -200
            n_img += max(1, int((ids[b] == -200).sum()))
        mask = None
        if B > 1:
            mask = np.ones((B, L), np.int64)
            for b in range(1, B):
                cut = int(rng.randint(1, L + 1))
                if cfg.tokenizer_padding_side == "left": mask[b, : L - cut] = 0
                else: mask[b, cut:] = 0
            n_img = sum(max(1, int(((ids[b] == -200) & (mask[b] == 1)).sum())) for b in range(B))
        pix = torch.from_numpy(synth.make_pixels(cfg, n_img, seed=1000 + case))
        ids_t = torch.from_numpy(ids); mask_t = None if mask is None else torch.from_numpy(mask)
        with torch.no_grad():
            ref = model(input_ids=ids_t, attention_mask=mask_t, images=pix, use_cache=True).logits.float()
            got = O.llava_forward(w, cfg, ids_t, pix, attention_mask=mask_t)[0]
        assert ref.shape == got.shape
        valid = torch.ones(ref.shape[:2], dtype=torch.bool)
        if mask is not None:                      # padded positions carry no contract (the reference's values there depend on the mask fill)
            with torch.no_grad():
                am = model.prepare_inputs_labels_for_multimodal(ids_t, None, mask_t, None, None, pix)[2]
            valid = am.bool()
        assert (ref - got)[valid].abs().max().item() <= 2e-5, (name, case)
        if B == 1:
            with torch.no_grad():
                gen = model.generate(inputs=ids_t, images=pix, do_sample=False, max_new_tokens=6, use_cache=True,
                                     past_key_values=ref_shim.subscriptable_cache())
                mine = O.greedy_generate(w, cfg, ids_t, pix, 6)
            assert gen[0, L:].tolist() == mine, (name, case)
