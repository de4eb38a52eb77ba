"""CPU, build container only: the oracle against the live REFERENCE at the REAL LLaVA-1.5-7B widths (hidden 4096, 32 heads of 128,
intermediate 11008, vocab 32000; CLIP ViT-L/14-336 widths, 577 tokens) with one decoder layer and one CLIP layer — the same
configuration tests/test_real_geometry_gpu.py checks the engine against the oracle with, so the chain
reference == oracle == engine is closed at full-size shapes too, not only on the tiny golden configs."""
import pytest
import torch

from oracle import llava_oracle as O, ref_shim
from synthetic import recipes as synth

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


@pytest.mark.parametrize("name,T", [("llava15_7b", 615), ("llava_plus_v0_7b", 295)])      # 336 px + mlp2x_gelu | 224 px + linear projector
def test_oracle_equals_reference_at_7b_widths(name, T):
    from dataclasses import replace
    cfg = replace(synth.with_layers(synth.CONFIGS[name], 1, 1), init="unit", mm_vision_select_layer=-1, max_position_embeddings=1024)
    wnp = synth.make_weights(cfg, 0)
    w = O.to_torch_weights(wnp)
    model = ref_shim.build_reference_model(cfg, wnp)
    ids = torch.from_numpy(synth.make_prompt(cfg, 40, image_positions=(17,), seed=5))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=6))
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        ref = model(input_ids=ids, images=pix, use_cache=True).logits.float()
        got = O.llava_forward(w, cfg, ids, pix)[0]
        feats_ref = model.encode_images(pix)
        feats = O.encode_images(w, cfg, pix)
    assert ref.shape == got.shape == (1, T, cfg.vocab_size)
    scale = ref.abs().max().item()
    assert (feats_ref - feats).abs().max().item() <= 2e-5 * max(1.0, feats_ref.abs().max().item())
    assert (ref - got).abs().max().item() <= 2e-5 * max(1.0, scale), ((ref - got).abs().max().item(), scale)
    with torch.no_grad():
        gen = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=4, use_cache=True, past_key_values=ref_shim.subscriptable_cache())
    assert gen[0, ids.shape[1]:].tolist() == O.greedy_generate(w, cfg, ids, pix, 4)
