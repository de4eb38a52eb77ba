"""Decode ms/token at LLaVA-1.5-7B geometry (1 image + 512-token prompt = 1087 positions) under several environment configurations, ONE process:
each argument is a comma-separated list of LMX_* switches ("LMX_DECODE_FLOW=0", "LMX_FLOW_R_O=1,LMX_FLOW_R_DOWN=1", "" = defaults).  The switches
are latched per model, so every configuration builds its own model (same seed -> same weights) and the generated ids of all configurations are
compared with the first one's (the flow kernel and the separate launches must agree bit for bit).

    python tools/mb_decode.py "" LMX_DECODE_FLOW=0 [--tokens 64] [--model llava15_7b] [--layers N]"""
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
        probe_ticks = None
        if os.environ.get("LMX_ATTN_PROBE") == "1":                       # of the last UNPROFILED token (event pairs stretch the gaps between launches)
            import ctypes
            tk = (ctypes.c_int64 * 64)(); nn = ctypes.c_int32(0)
            _C.check(_C.lib.lmx_flow_timeline(model._h, tk, 64, ctypes.byref(nn)))
            probe_ticks = [x / 100.0 for x in tk[:32]]
        model.profile(True)
        _C.check(_C.lib.lmx_decode(model._h, c.seqs[0], -1, 16, None, 1, _C.stream_handle()))
        prof = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in model.profile_read().items() if k.startswith("decode.")}       # us per launch
        model.profile(False)
        timeline = None
        if probe_ticks is not None:
            t = probe_ticks
            timeline = {"attn block(head0,split0) us since its start [small loads, K landed, V landed, partial stored, LAST block's partial stored]": [round(t[k] - t[0], 2) for k in (1, 2, 3, 4, 5)],
                        "merger(head0) us since block0 start [start, small, K, V, partial, poll ok, out stored]": [round(t[8 + k] - t[0], 2) for k in range(7)],
                        "o_proj wg0 us since attn block0 start [loads issued, tails landed, barrier, merged, barrier, stream done]": [round(t[16 + k] - t[0], 2) for k in range(6)],
                        "o_proj wg128": [round(t[24 + k] - t[0], 2) for k in range(6)], "o_proj last wg stream done": round(t[15] - t[0], 2)}
        if os.environ.get("LMX_FLOW_TIMELINE") == "1" and os.environ.get("LMX_DECODE_ENGINE") == "1":
            import ctypes
            L = cfg.num_hidden_layers
            ns = 5 * L + 1
            tk = (ctypes.c_int64 * (3 * ns))(); nn = ctypes.c_int32(0)
            _C.check(_C.lib.lmx_flow_timeline(model._h, tk, 3 * ns, ctypes.byref(nn)))
            t = [x / 100.0 for x in tk[: 3 * ns]]
            names = ("qkv", "attn", "o", "gate_up", "down")
            def mean(f):
                return {k: round(sum(f(l * 5 + j) for l in range(1, L)) / max(L - 1, 1), 2) for j, k in enumerate(names)}
            timeline = {"total_us": round(t[3 * (ns - 1) + 2] - t[0], 1),
                        "gather_us(input gathered - step entered)": mean(lambda s_: t[3 * s_ + 1] - t[3 * s_]),
                        "stream_us(done - gathered)": mean(lambda s_: t[3 * s_ + 2] - t[3 * s_ + 1]),
                        "step_us(entered next - entered)": mean(lambda s_: t[3 * (s_ + 1)] - t[3 * s_])}
        elif os.environ.get("LMX_FLOW_TIMELINE") == "1":
            import ctypes
            L = cfg.num_hidden_layers
            ns = 5 * L + 1
            tk = (ctypes.c_int64 * (5 * ns + 1))(); nn = ctypes.c_int32(0)
            _C.check(_C.lib.lmx_flow_timeline(model._h, tk, 5 * ns + 1, ctypes.byref(nn)))
            if nn.value:
                t = [x / 100.0 for x in tk[: nn.value]]                  # us
                done, go, xs, go_l, xs_l = (t[1 + i * ns:1 + (i + 1) * ns] for i in range(5))
                names = ("qkv", "attn", "o", "gate_up", "down")
                def mean(f):
                    return {k: round(sum(f(l * 5 + j) for l in range(1, L)) / max(L - 1, 1), 2) for j, k in enumerate(names)}
                timeline = {"total_us": round(done[-1] - t[0], 1),
                            "step_us(done - prev done)": mean(lambda s_: done[s_] - done[s_ - 1]),
                            "release_us(first past wait - prev done)": mean(lambda s_: go[s_] - done[s_ - 1]),
                            "stage_us(first x staged - first past wait)": mean(lambda s_: xs[s_] - go[s_]),
                            "work_us(done - first x staged)": mean(lambda s_: done[s_] - xs[s_]),
                            "last_release_us(last past wait - prev done)": mean(lambda s_: go_l[s_] - done[s_ - 1]),
                            "last_stage_us(last x staged - prev done)": mean(lambda s_: xs_l[s_] - done[s_ - 1]),
                            "lm_head_us": round(done[-1] - done[-2], 2)}
        if first is None:
            first = got
        same = got == first
        print(json.dumps({"kind": "decode_step", "config": conf or "(defaults)", "us_per_token": [round(t, 1) for t in times], "best_us": round(min(times), 1),
                          "context_end": int(embeds.shape[1]) + len(got), "n_ids": len(got), "ids_equal_first": same, "ids_head": got[:6], "ids_hash": __import__("hashlib").sha1(str(got).encode()).hexdigest()[:12], "timeline": timeline, "us_per_launch": prof}), flush=True)
        del c, model
        gc.collect(); torch.cuda.empty_cache()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
