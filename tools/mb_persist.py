"""Per-launch time of the persistent decode-step kernel at LLaVA-1.5-7B geometry, context 1087 (env: LMX_DECODE_PERSIST_STEPS = first k steps only,
LMX_DECODE_PERSIST_GRID, LMX_DECODE_PERSIST_FENCE, LMX_DECODE_PERSIST=0 for the separate launches)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C
from llava_mi355x.model import LmxKVCache
from synthetic import build as harness, recipes as synth
cfg = synth.CONFIGS["llava15_7b"]
dev = torch.device("cuda:0")
model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, max_position=2048)
ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None].to(dev)
pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, torch.bfloat16)
_, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
c = LmxKVCache(model, 1)
_C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
_C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, 8, None, 1, _C.stream_handle()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, n, None, 1, _C.stream_handle()))
e1.record(); torch.cuda.synchronize()
print(json.dumps({"kind": "persist_step", "env": {k: v for k, v in os.environ.items() if k.startswith("LMX_DECODE")}, "us_per_token": e0.elapsed_time(e1) / n * 1e3}), flush=True)
