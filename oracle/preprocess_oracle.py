"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the image preprocessing in front of the vision tower:
`process_images` / `expand2square` (llava/mm_utils.py:16-44) feeding HF's CLIPImageProcessor (resize shortest edge with PIL
BICUBIC -> center crop -> rescale 1/255 -> normalise), as numpy.

The resize is Pillow's two-pass 8-bit resampler restated from its published algorithm (Pillow, src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc): double-precision bicubic (a = -0.5)
coefficients over a support scaled by the downsampling factor, normalised, converted to 22-bit fixed point, integer
accumulation with round-half-up and clipping to uint8 after EACH pass (horizontal first).  Pinned bit-exactly against
PIL.Image.resize itself and to <= 1e-6 against the installed CLIPImageProcessor in tests/test_preprocess.py (CPU).
Only tests / smoke / the bench's cpu leg may import this module."""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (bounds [out,2] int32 (first source index, count), coeffs [out, ksize] int32 fixed point, ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """img [H, W, C] uint8 -> [out_h, out_w, C] uint8, bit-exact with PIL.Image.resize((out_w, out_h), BICUBIC)."""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        bounds, kk, _ = precompute_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1)
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if out_h != H:
        bounds, kk, _ = precompute_coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], C), np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = (1 << (PRECISION_BITS - 1)) + (src[y0:y0 + n, :, :] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc)
        cur = out
    return cur


def expand2square_u8(img: np.ndarray, fill: Tuple[int, int, int]) -> np.ndarray:
    """llava/mm_utils.py:16-27: centre the image on a square canvas of the fill colour."""
    H, W, C = img.shape
    if H == W:
        return img
    S = max(H, W)
    out = np.empty((S, S, C), np.uint8)
    out[:] = np.asarray(fill, np.uint8)
    y0, x0 = (S - H) // 2, (S - W) // 2
    out[y0:y0 + H, x0:x0 + W] = img
    return out


def resize_output_size(H: int, W: int, short: int) -> Tuple[int, int]:
    """HF get_resize_output_image_size(size=short, default_to_square=False): (new_h, new_w)."""
    s, l = (W, H) if W <= H else (H, W)
    new_s, new_l = short, int(short * l / s)
    return (new_l, new_s) if W <= H else (new_s, new_l)


def clip_preprocess(img: np.ndarray, size: int = 336, pad: bool = False, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """uint8 RGB [H, W, 3] -> float32 [3, size, size]  (process_images + CLIPImageProcessor)."""
    if pad:
        img = expand2square_u8(img, tuple(int(c * 255) for c in mean))
    H, W, _ = img.shape
    nh, nw = resize_output_size(H, W, size)
    r = resize_bicubic_u8(img, nw, nh)
    top, left = (nh - size) // 2, (nw - size) // 2
    c = r[top:top + size, left:left + size].astype(np.float32)
    out = (c * np.float32(1.0 / 255.0) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(out.transpose(2, 0, 1))
