"""Decode-batch microbenchmark at real geometry: ms/step and aggregate tokens/s for B sequences stepping together
(lmx_decode_batch) vs B=1 through the single-sequence path.  Usage: python tools/mb_batch.py [model] [ctx_prompt_len]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))


def main():
    from llava_mi355x import _C
    from llava_mi355x.batching import DecodeBatch
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness, recipes as synth
    name = sys.argv[1] if len(sys.argv) > 1 else "llava15_7b"
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    sizes = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,4,8,16,32".split(","))]
    cfg = synth.CONFIGS[name]
    dev = torch.device("cuda:0")
    model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, max_position=2048)
    ids = torch.from_numpy(synth.make_prompt(cfg, L, image_positions=(35,), seed=2))[None].to(dev)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, torch.bfloat16)
    _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
    T = embeds.shape[1]
    maxb = max(sizes)
    caches = []
    for _ in range(maxb):
        c = LmxKVCache(model, 1)
        _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), T, 0, None, 0, 1, _C.stream_handle()))
        caches.append(c)
    torch.cuda.synchronize()
    steps = 24
    res = []
    # single-sequence path
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _C.check(_C.lib.lmx_decode(model._h, caches[0].seqs[0], -1, 4, None, 1, _C.stream_handle()))
    e0.record(); _C.check(_C.lib.lmx_decode(model._h, caches[0].seqs[0], -1, steps, None, 1, _C.stream_handle())); e1.record()
    torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1) / steps
    res.append({"path": "single (gemv)", "B": 1, "ms_per_step": ms1, "tokens_per_s": 1e3 / ms1})
    batch = DecodeBatch(model, maxb)
    for B in sizes:
        seqs = [c.seqs[0] for c in caches[:B]]
        batch.step(seqs, None, 3, True, want_ids=False)
        torch.cuda.synchronize()
        e0.record(); batch.step(seqs, None, steps, True, want_ids=False); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        res.append({"path": "batch", "B": B, "ms_per_step": ms, "tokens_per_s": B * 1e3 / ms})
    # per-kernel split of one batched step at the largest B
    model.profile(True)
    batch.step([c.seqs[0] for c in caches[:maxb]], None, 2, True, want_ids=False)
    prof = model.profile_read(); model.profile(False)
    for r in res:
        print(json.dumps(r))
    print(json.dumps({"profile_B": maxb, "ctx": T, "per_step_ms": {k: round(v[0] / 2, 4) for k, v in sorted(prof.items()) if k.startswith("decode_batch")}}))


if __name__ == "__main__":
    main()
