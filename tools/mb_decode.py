"""Decode ms/token at LLaVA-1.5-7B geometry (1 image + 512-token prompt = 1087 positions) under several environment configurations, ONE process:
each argument is a comma-separated list of LMX_* switches ("LMX_DECODE_SPLITQ=0", "" = defaults).  The switches are read when a model is created, so
every configuration builds its own model (same seed -> same weights) and the generated ids of all configurations are compared with the first one's
(the split-q decode step and the three-launch form must agree bit for bit).

    python tools/mb_decode.py "" LMX_DECODE_SPLITQ=0 [--tokens 64] [--model llava15_7b] [--layers N]"""
import argparse, gc, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C
from llava_mi355x.model import LmxKVCache
from synthetic import build as harness, recipes as synth

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=[""])
ap.add_argument("--tokens", type=int, default=64)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--model", default="llava15_7b")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--prompt", type=int, default=512)
args = ap.parse_args()

cfg = synth.CONFIGS[args.model]
if args.layers:
    cfg = synth.with_layers(cfg, args.layers, 1)
dev = torch.device("cuda:0")
ids = torch.from_numpy(synth.make_prompt(cfg, args.prompt, image_positions=(35,), seed=2))[None].to(dev)
pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, torch.bfloat16)
first = None
for conf in args.configs:
    env = dict(kv.split("=", 1) for kv in conf.replace(";", " ").split() if kv) if (";" in conf or " " in conf) else dict(kv.split("=", 1) for kv in conf.split(",") if kv)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        model = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, max_position=2048)
        _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
        c = LmxKVCache(model, 1)
        _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
        _C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, 8, None, 1, _C.stream_handle()))
        torch.cuda.synchronize()
        times = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, args.tokens, None, 1, _C.stream_handle()))
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / args.tokens * 1e3)
        got = c.generated_ids(0) if hasattr(c, "generated_ids") else None
        if got is None:
            import ctypes
            n = 8 + args.rounds * args.tokens + 1
            buf = torch.zeros(n, dtype=torch.int64)
            cnt = ctypes.c_int32(0)
            _C.check(_C.lib.lmx_seq_read_tokens(c.seqs[0], _C.ptr(buf), n, ctypes.byref(cnt), _C.stream_handle()))
            got = buf[:cnt.value].tolist()
        model.profile(True)
        _C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, 8, None, 1, _C.stream_handle()))
        torch.cuda.synchronize()
        prof = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in model.profile_read().items() if k.startswith("decode")}
        model.profile(False)
        if first is None:
            first = got
        same = got == first
        print(json.dumps({"kind": "decode_step", "config": conf or "(defaults)", "us_per_token": [round(t, 1) for t in times], "best_us": round(min(times), 1),
                          "context_end": int(embeds.shape[1]) + len(got), "n_ids": len(got), "ids_equal_first": same, "ids_head": got[:6], "ids_hash": __import__("hashlib").sha1(str(got).encode()).hexdigest()[:12], "us_per_launch": prof}), flush=True)
        del c, model
        gc.collect(); torch.cuda.empty_cache()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
