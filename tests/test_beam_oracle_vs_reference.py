"""Pins oracle/beam_oracle.py to the reference: the unmodified LlavaLlamaForCausalLM of /root/reference (through oracle/ref_shim.py) runs its own
`generate(num_beams=k)` — the call the reference's eval scripts make (llava/eval/run_llava.py:121) — and the oracle's restatement must return the same
ids.  Cases without EOS (fixed length: every transformers release agrees on them).  Skipped where /root/reference is absent (GPU box)."""
import numpy as np
import pytest
import torch

from oracle import beam_oracle, llava_oracle as O, ref_shim
from synthetic import recipes as synth

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")


@pytest.mark.parametrize("name,beams,new", [("tiny", 3, 8), ("tiny", 2, 6), ("tiny_gqa", 4, 6)])
def test_beam_oracle_matches_reference_generate(name, beams, new):
    cfg = synth.CONFIGS[name]
    wnp = synth.make_weights(cfg, 0)
    model = ref_shim.build_reference_model(cfg, wnp)
    ids = torch.from_numpy(synth.make_prompt(cfg, 20, image_positions=(5,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=3))
    with torch.no_grad():
        ref = model.generate(inputs=ids, images=pix, do_sample=False, num_beams=beams, max_new_tokens=new, use_cache=True,
                             past_key_values=ref_shim.subscriptable_cache())
    want = ref[0, ids.shape[1]:].tolist()
    got = beam_oracle.beam_search(O.to_torch_weights(wnp), cfg, ids, pix, beams, new)
    assert got == want
    greedy = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=new, use_cache=True, past_key_values=ref_shim.subscriptable_cache())
    print(name, beams, "beam", want, "greedy", greedy[0, ids.shape[1]:].tolist())
