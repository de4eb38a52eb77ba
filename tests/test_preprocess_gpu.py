"""Device preprocessing (csrc/preprocess.hip) vs the oracle (pinned to PIL / the HF processor in tests/test_preprocess.py):
identical fp32 pixel values — the uint8 stage is integer arithmetic, the float tail uses the same three fp32 operations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _img(seed, w, h):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    if seed % 2:
        base = (base.astype(np.float32) * 0.3 + np.linspace(0, 180, w, dtype=np.float32)[None, :, None]).clip(0, 255).astype(np.uint8)
    return base


class _Proc:
    image_mean = [0.48145466, 0.4578275, 0.40821073]
    image_std = [0.26862954, 0.26130258, 0.27577711]

    def __init__(self, S):
        self.size = {"shortest_edge": S}; self.crop_size = {"height": S, "width": S}


class _Cfg:
    def __init__(self, pad):
        self.image_aspect_ratio = "pad" if pad else None


@pytest.mark.parametrize("w,h", [(640, 480), (50, 70), (56, 56), (70, 50), (1024, 300), (33, 57), (2000, 1500), (56, 90)])
@pytest.mark.parametrize("pad", [False, True])
def test_device_preprocess_equals_oracle(cuda, w, h, pad):
    from PIL import Image
    from llava_mi355x.mm_utils import process_images_device
    from synthetic import build as harness
    from synthetic import recipes as synth
    from oracle.preprocess_oracle import clip_preprocess
    cfg = synth.CONFIGS["tiny"]                      # tower image size 56
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    S = cfg.v_image_size
    img = _img(w * 3 + h + int(pad), w, h)
    got = process_images_device([Image.fromarray(img, "RGB")], _Proc(S), _Cfg(pad), model)[0].cpu().numpy()
    ref = clip_preprocess(img, S, pad)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_device_preprocess_real_size_and_worker_hook(cuda, monkeypatch):
    """336-px tower geometry (one CLIP layer so the model is small): equality with the oracle at the real output size, and
    process_images() switches to the device path through the environment variable the worker would set."""
    from dataclasses import replace
    from PIL import Image
    from llava_mi355x import mm_utils
    from synthetic import build as harness
    from synthetic import recipes as synth
    from oracle.preprocess_oracle import clip_preprocess
    cfg = replace(synth.CONFIGS["tiny"], name="tiny336", v_image_size=336, max_position_embeddings=1024)
    model = harness.build_model(cfg, dtype=torch.float32, seed=0)
    imgs = [_img(11, 640, 427), _img(12, 500, 800)]
    pils = [Image.fromarray(i, "RGB") for i in imgs]
    mm_utils.set_device_preprocess_model(model)
    monkeypatch.setenv("LLAVA_MI355X_DEVICE_PREPROCESS", "1")
    for pad in (False, True):
        out = mm_utils.process_images(pils, _Proc(336), _Cfg(pad))
        assert out.is_cuda and out.shape == (2, 3, 336, 336)
        for i in range(2):
            assert np.array_equal(out[i].cpu().numpy(), clip_preprocess(imgs[i], 336, pad))
    mm_utils.set_device_preprocess_model(None)
