"""How much of the tensor-parallel prefill all-reduce does the two-half pipeline hide?  

ONE engine instance holds rank 0's shard of a TP=W group (LLaVA-1.5-7B geometry: the rank-local GEMM / attention shapes are the production ones) and the
all-reduce is replaced through lmx_tp_set_allreduce_hook by a stand-in that only TAKES TIME on the stream it is given: torch.cuda._sleep for
bytes / link_GBps + latency (no partner needed; an RCCL all-reduce kernel also occupies few CUs).  Prefill of the config-2 request (1087 positions) is
timed with no communication cost, serialised (LMX_TP_OVERLAP=0: every all-reduce on the launch stream) and pipelined (row halves, comm stream).
    hidden fraction = (t_serial - t_overlap) / (t_serial - t_nocomm)
usage: mb_tp_overlap.py [world=2] [link_GBps=153] [latency_us=12] [prompt_len=512]"""
import ctypes, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import _C
from llava_mi355x.model import LmxKVCache
from synthetic import build as harness, recipes as synth

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
link = float(sys.argv[2]) if len(sys.argv) > 2 else 153.0
lat_us = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
prompt_len = int(sys.argv[4]) if len(sys.argv) > 4 else 512
HOOK_T = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)
dev = torch.device("cuda:0")
cfg = synth.CONFIGS["llava15_7b"]

# cycles of torch.cuda._sleep per microsecond
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
state = {"on": True, "calls": 0, "us": 0.0}

def hook(buf, count, dtype_code, stream, ctx):
    state["calls"] += 1
    if not state["on"]:
        return
    # ring all-reduce between W ranks: 2 (W - 1) / W of the message crosses a link at link GB/s, plus a fixed latency
    us = lat_us + 2.0 * (world - 1) / world * int(count) * 2 / (link * 1e3)
    state["us"] += us
    sp = int(stream) if stream else 0
    strm = torch.cuda.ExternalStream(sp) if sp else torch.cuda.default_stream(dev)
    with torch.cuda.stream(strm):
        torch.cuda._sleep(int(us * cyc_per_us))

def build(overlap):
    os.environ["LMX_TP_OVERLAP"] = "1" if overlap else "0"
    m = harness.build_model(cfg, dtype=torch.bfloat16, seed=0, device_rng=True, device=dev, tp_rank=0, tp_world=world, max_position=4096)
    h = HOOK_T(hook); m._hook_keepalive = h
    _C.check(_C.lib.lmx_tp_set_allreduce_hook(m._h, ctypes.cast(h, ctypes.c_void_p), None))
    return m

ids = torch.from_numpy(synth.make_prompt(cfg, prompt_len, image_positions=(35,), seed=2))[None].to(dev)
pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, torch.bfloat16)

def time_prefill(m, comm_on, reps=5):
    state["on"] = comm_on
    _, _, _, _, embeds, _ = m.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
    ts = []
    for r in range(reps + 1):
        c = LmxKVCache(m, 1)
        state["calls"] = 0; state["us"] = 0.0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _C.check(_C.lib.lmx_prefill(m._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
        b.record(); torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
        c.close()
    return sorted(ts)[len(ts) // 2], state["calls"], state["us"]

res = {"kind": "tp_overlap", "world": world, "link_GBps": link, "latency_us": lat_us, "positions": int(ids.shape[1] - 1 + cfg.tokens_per_image)}
ms, mo = build(False), None
t_nocomm, _, _ = time_prefill(ms, False)
t_serial, calls_s, comm_us = time_prefill(ms, True)
del ms; torch.cuda.empty_cache()
mo = build(True)
t_nocomm_o, _, _ = time_prefill(mo, False)
t_overlap, calls_o, comm_us_o = time_prefill(mo, True)
res.update({"prefill_ms_no_comm_cost": t_nocomm, "prefill_ms_no_comm_cost_two_halves": t_nocomm_o, "prefill_ms_serialised": t_serial, "prefill_ms_overlapped": t_overlap,
            "allreduce_calls": [calls_s, calls_o], "modelled_comm_ms": [comm_us / 1e3, comm_us_o / 1e3],
            "hidden_fraction": (t_serial - t_overlap) / max(t_serial - t_nocomm, 1e-9),
            "note": "decoder layers of rank 0's shard only (tower + projector are replicated and not in this timing)"})
print(json.dumps(res), flush=True)
