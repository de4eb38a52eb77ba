"""CPU: the preprocessing oracle (oracle/preprocess_oracle.py) pinned against the libraries the reference calls —
PIL.Image.resize(BICUBIC) bit for bit, expand2square (llava/mm_utils.py:16-27) and the installed CLIPImageProcessor."""
import numpy as np
import pytest


def _img(seed, w, h):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    # smooth half of the cases a little so both noisy and natural-ish content is covered
    if seed % 2:
        base = (base.astype(np.float32) * 0.3 + np.linspace(0, 180, w, dtype=np.float32)[None, :, None]).clip(0, 255).astype(np.uint8)
    return base


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 448, 336), (50, 70, 336, 470), (336, 336, 336, 336), (1000, 333, 1009, 336),
                                       (37, 41, 336, 372), (800, 600, 224, 168), (336, 500, 336, 500), (123, 77, 61, 39)])
def test_resize_bit_exact_with_pil(w, h, ow, oh):
    from PIL import Image
    from oracle.preprocess_oracle import resize_bicubic_u8
    img = _img(w + h, w, h)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), resample=Image.BICUBIC))
    got = resize_bicubic_u8(img, ow, oh)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("w,h", [(640, 480), (50, 70), (336, 336), (70, 50), (1024, 300)])
@pytest.mark.parametrize("pad", [False, True])
def test_clip_preprocess_matches_hf_processor(w, h, pad):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
    from PIL import Image
    from transformers import CLIPImageProcessor
    from llava_mi355x.mm_utils import expand2square
    from oracle.preprocess_oracle import clip_preprocess, expand2square_u8
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    img = _img(3 * w + h, w, h)
    pil = Image.fromarray(img, "RGB")
    if pad:
        fill = tuple(int(c * 255) for c in proc.image_mean)
        sq = expand2square(pil, fill)
        assert np.array_equal(np.asarray(sq), expand2square_u8(img, fill))
        pil = sq
    ref = proc.preprocess(pil, return_tensors="np")["pixel_values"][0]
    got = clip_preprocess(img, 336, pad)
    assert got.shape == ref.shape == (3, 336, 336)
    assert np.abs(got - ref).max() <= 1e-6


@pytest.mark.parametrize("in_size,out_size", [(640, 448), (50, 336), (336, 336), (1000, 1009), (37, 336), (800, 224), (4000, 336), (123, 61)])
def test_library_resample_tables_equal_oracle(in_size, out_size):
    """Host half of lmx_preprocess_image (no GPU needed): the coefficient tables the library uploads are, entry for entry, the
    oracle's — i.e. Pillow's precompute_coeffs + normalize_coeffs_8bpc (the oracle is pinned bit-exactly to PIL above)."""
    import ctypes, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
    from llava_mi355x import _C
    from oracle.preprocess_oracle import precompute_coeffs
    bounds, kk, ksize = precompute_coeffs(in_size, out_size)
    for first, n in ((0, out_size), (out_size // 3, out_size - out_size // 3)):
        b = (ctypes.c_int32 * (2 * n))(); k = (ctypes.c_int32 * (n * ksize))()
        got = _C.lib.lmx_preprocess_coeffs(in_size, out_size, first, n, b, k, n * ksize)
        assert got == ksize
        assert np.array_equal(np.frombuffer(b, np.int32).reshape(n, 2), bounds[first:first + n])
        assert np.array_equal(np.frombuffer(k, np.int32).reshape(n, ksize), kk[first:first + n])
