"""Shared helpers for golden-vector tests (tests/golden/*.npz, produced by oracle/make_golden.py from the reference)."""
import json
import os
from dataclasses import replace

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def case_inputs(z, meta, cname):
    from synthetic import recipes as synth
    base = synth.CONFIGS[meta["config"]]
    cm = meta["cases"][cname]
    cfg = replace(base, **cm["cfg"])
    p = cname + "."
    ids = z[p + "input_ids"]
    mask = z[p + "attention_mask_in"] if p + "attention_mask_in" in z.files else None
    labels = z[p + "labels_in"] if p + "labels_in" in z.files else None
    pix = synth.make_pixels(cfg, cm["n_images"], seed=1)
    return cfg, cm, ids, mask, labels, pix


def split_images(pix, as_list):
    if not as_list:
        return pix
    out, o = [], 0
    for n in as_list:
        out.append(pix[o:o + n]); o += n
    return out
