"""Pins the training-step oracle (torch autograd over oracle/llava_oracle.py + the HF shifted cross-entropy, what tests/test_train_step_gpu.py compares the
HIP step with) to the reference's own code: the unmodified LlavaLlamaForCausalLM of /root/reference (oracle/ref_shim.py) runs forward(labels=...) —
llava_llama.py:56-99, the call HF Trainer.training_step makes under llava/train/train.py:805-1000 — and loss.backward(); loss and the gradients of every
trainable tensor (LLM + mm_projector; the tower is frozen, clip_encoder.py:25) must equal the oracle's.  Skipped where /root/reference is absent."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import llava_oracle as O, ref_shim
from synthetic import recipes as synth

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
def test_oracle_loss_and_gradients_equal_the_reference_models(name):
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    model = ref_shim.build_reference_model(cfg, wnp)
    model.train()
    a = synth.make_prompt(cfg, 22, image_positions=(4,), seed=2); b = synth.make_prompt(cfg, 22, image_positions=(7,), seed=3)
    ids = torch.from_numpy(np.stack([a, b]))
    mask = torch.ones_like(ids); mask[1, 15:] = 0
    labels = ids.clone(); labels[0, :9] = -100; labels[1, :11] = -100; labels[ids == -200] = -100
    pix = torch.from_numpy(synth.make_pixels(cfg, 2, seed=1))
    for p in model.parameters():
        p.requires_grad_(True)
    for p in model.get_vision_tower().parameters():
        p.requires_grad_(False)
    out = model(input_ids=ids, attention_mask=mask, labels=labels, images=pix)
    out.loss.backward()
    ref_grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}

    w = O.to_torch_weights(wnp)
    params = {k: v.clone().requires_grad_(True) for k, v in w.items() if not k.startswith("vision.") and "vision_tower" not in k}
    full = dict(w); full.update(params)
    logits, _, _, new_labels = O.llava_forward(full, cfg, ids, pix, attention_mask=mask, labels=labels)
    loss = F.cross_entropy(logits[:, :-1].reshape(-1, cfg.vocab_size), new_labels[:, 1:].reshape(-1), ignore_index=-100)
    loss.backward()
    assert abs(loss.item() - out.loss.item()) <= 1e-5 * abs(out.loss.item())
    checked = 0
    for k, p in params.items():
        cands = [rk for rk in ref_grads if rk.endswith(k) or rk.endswith(k.replace("mm_projector.", "model.mm_projector."))]
        if not cands:
            continue
        g = ref_grads[cands[0]]
        assert g.shape == p.grad.shape, k
        assert (p.grad - g).abs().max().item() <= 2e-5 * max(g.abs().max().item(), 1e-6) + 1e-8, k
        checked += 1
    assert checked >= len(params) - 1, (checked, len(params))
