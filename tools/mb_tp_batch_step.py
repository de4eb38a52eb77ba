"""Rank-local batched decode step of a tensor-parallel shard on ONE GPU (all-reduces replaced by a C no-op): ms per step of `batch` sequences at the headline context.
usage: mb_tp_batch_step.py <world> <batch> [steps] [max_position]   (
max_position = the KV-cache capacity = the row pitch of the V^T cache in keys: 2048 by default, e.g. 2176 for a pitch that is not a power of two)"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd")); sys.path.insert(0, ROOT)


def main():
    W, B = int(sys.argv[1]), int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    max_pos = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
    from llava_mi355x import _C
    from llava_mi355x.batching import DecodeBatch
    from llava_mi355x.model import LmxKVCache
    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS["llava15_7b"]
    dev = torch.device("cuda:0")
    m = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, tp_rank=0, tp_world=W, max_position=max_pos)
    _C.check(_C.lib.lmx_tp_set_allreduce_hook(m._h, ctypes.cast(ctypes.CDLL(None).getpid, ctypes.c_void_p), None))
    ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None].to(dev)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, torch.bfloat16)
    _, _, _, _, embeds, _ = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
    caches = []
    for _ in range(B):
        c = LmxKVCache(m, 1)
        _C.check(_C.lib.lmx_prefill(m._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
        caches.append(c)
    bt = DecodeBatch(m, B)
    seqs = [c.seqs[0] for c in caches]
    bt.step(seqs, None, 2, True, want_ids=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); bt.step(seqs, None, steps, True, want_ids=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps)
    m.profile(True)
    bt.step(seqs, None, 8, True, want_ids=False)
    torch.cuda.synchronize()
    prof = {k: [round(v[0] / max(v[1], 1) * 1e3, 2), int(v[1] / 8)] for k, v in m.profile_read().items() if k.startswith("decode_batch")}
    print(json.dumps({"world": W, "batch": B, "kv_capacity": max_pos, "ms_per_step": round(sorted(ts)[1], 4), "us_per_launch_and_launches_per_step": prof}), flush=True)


if __name__ == "__main__":
    main()
