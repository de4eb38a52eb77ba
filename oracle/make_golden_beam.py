"""TEST INFRASTRUCTURE — writes tests/golden/beam.npz: ids returned by the REFERENCE model's own `generate(num_beams=k)` (llava/eval/run_llava.py:121
passes num_beams through; model built by oracle/ref_shim.py from the unmodified /root/reference sources) on seeded tiny requests.  Inputs are
regenerated from synthetic/recipes.py (config, prompt length / marker position / seeds below), so the file holds outputs only.
    python -m oracle.make_golden_beam"""
import json
import os

import numpy as np
import torch

from oracle import ref_shim
from synthetic import recipes as synth

CASES = [("tiny", 3, 8, 20, 5, 2, 3), ("tiny", 2, 6, 20, 5, 2, 3), ("tiny", 4, 10, 26, 9, 7, 8), ("tiny_gqa", 4, 6, 20, 5, 2, 3), ("tiny_gqa", 3, 9, 24, 3, 11, 12)]
# (config, num_beams, max_new_tokens, prompt length, image marker position, prompt seed, pixel seed)


def main():
    out, meta = {}, []
    for i, (name, beams, new, L, pos, s_ids, s_pix) in enumerate(CASES):
        cfg = synth.CONFIGS[name]
        model = ref_shim.build_reference_model(cfg, synth.make_weights(cfg, 0))
        ids = torch.from_numpy(synth.make_prompt(cfg, L, image_positions=(pos,), seed=s_ids))[None]
        pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=s_pix))
        with torch.no_grad():
            ref = model.generate(inputs=ids, images=pix, do_sample=False, num_beams=beams, max_new_tokens=new, use_cache=True,
                                 past_key_values=ref_shim.subscriptable_cache())
            greedy = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=new, use_cache=True, past_key_values=ref_shim.subscriptable_cache())
        out[f"case{i}.beam"] = ref[0, L:].numpy()
        out[f"case{i}.greedy"] = greedy[0, L:].numpy()
        meta.append(dict(config=name, num_beams=beams, max_new_tokens=new, prompt_len=L, image_pos=pos, seed_ids=s_ids, seed_pix=s_pix))
    out["meta"] = np.frombuffer(json.dumps({"cases": meta, "transformers": ref_shim.load_reference().transformers_version}).encode(), dtype=np.uint8)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "beam.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, [(m["config"], m["num_beams"], out[f"case{i}.beam"].tolist(), out[f"case{i}.greedy"].tolist()) for i, m in enumerate(meta)])


if __name__ == "__main__":
    main()
