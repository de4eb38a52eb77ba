"""CLIP tower fusions of round 4: the K rows / V^T columns of a layer leave the q|k|v GEMM's epilogue (GemmArgs::pk_*; gemm_common.h) instead of a pack launch
per layer and image.  Same bias, same rounding, same bytes in the caches: encode_images must not change by a bit (HF5:models/clip/modeling_clip.py:295-340)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dtype,n_img", [("tiny", torch.bfloat16, 3), ("tiny", torch.float16, 1), ("llava15_7b", torch.bfloat16, 2)])
def test_kv_pack_in_the_qkv_epilogue_is_bit_identical(cuda, name, dtype, n_img):
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[name]
    if name != "tiny":
        cfg = synth.with_layers(cfg, 1, 3)                     # real ViT-L/14-336 widths (577 rows, 16 heads x 64), three tower layers
    model = harness.build_model(cfg, dtype=dtype, seed=0, device_rng=name != "tiny", **({} if name == "tiny" else {"max_position": 2048}))
    pix = torch.from_numpy(synth.make_pixels(cfg, n_img, seed=11)).to(cuda, dtype)
    try:
        model.set_option("vis_pack", 0)
        model.profile(True); ref = model.encode_images(pix); names0 = model.profile_read(); model.profile(False)
        model.set_option("vis_pack", 1)
        model.profile(True); got = model.encode_images(pix); names1 = model.profile_read(); model.profile(False)
    finally:
        model.set_option("vis_pack", 1)
    assert torch.equal(got, ref)
    assert names0.get("vis.kv_pack", (0.0, 0))[1] > 0 and names1.get("vis.kv_pack", (0.0, 0))[1] == 0      # the pack launches are gone
