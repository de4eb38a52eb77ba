import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/llava-plus-codebase_amd")
from synthetic import build as harness, recipes as synth
cfg = synth.CONFIGS["tiny"]
m = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, weights=synth.make_weights(cfg, 0))
ids = torch.tensor([[1, 5, 7, 9]], device="cuda")
out = m.generate(inputs=ids, do_sample=False, max_new_tokens=3, eos_token_id=-1)
torch.cuda.synchronize()
print("ok", out.tolist(), flush=True)
