"""Timeline of the split-q decode step's second launch (csrc/decode_attn.hip: decode_kv_attn_kernel) at the headline geometry, stand-alone.

Rotates over 32 "layers" of weights and KV caches (nothing is served by a cache the real step would not have), launches the k|v projection + attention launch per
layer and prints, per launch, event time and the in-kernel clock stamps (DecAttnArgs::ts) relative to the earliest workgroup start, in microseconds:
  attn0 / proj0      earliest attention / projection workgroup start
  partial            head 0's merger: own partial reduced          arrivals: the other chunks' arrivals seen        merged: partials merged
  granules           k_new | v_new granules of the head arrived    stored: output row stored
  proj_end           latest projection row published               last_merger / last_chunk: latest merger / non-merger workgroup done
usage: mb_kv_attn.py [pos] [layers]"""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
from llava_mi355x import ops  # noqa: E402


def main():
    pos = int(sys.argv[1]) if len(sys.argv) > 1 else 1150
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    nh = nkv = 32; D = 128; K = 4096; s_max = 2048
    T = torch.bfloat16
    torch.manual_seed(0)
    w = (torch.randn(L, (nh + 2 * nkv) * D, K, device=dev) / math.sqrt(K)).to(T)
    kc = torch.randn(L, nkv, s_max, D, device=dev).to(T)
    vt = torch.randn(L, nkv, D, s_max, device=dev).to(T)
    table = torch.randn(s_max, D, device=dev)
    g = torch.ones(K, device=dev, dtype=T)
    x = torch.randn(K, device=dev).to(T)
    q_row = torch.randn((nh + 2 * nkv) * D, device=dev).to(T)
    gran = torch.zeros(2 * nkv * D, dtype=torch.int64, device=dev)
    big = 0x7fffffffffffffff
    tl = torch.zeros(L, 10, dtype=torch.int64, device=dev)
    tag = [1]
    scratch = ops.decode_attn_scratch(nh, s_max // 128, D, T, dev)
    y = torch.empty(1, 2 * nkv * D, device=dev, dtype=T)

    def run(with_tl):
        for l in range(L):
            tag[0] += 1
            ops.decode_kv_attn(q_row, x, w[l, nh * D:], g, 1e-5, kc[l], vt[l], table, pos, nh, nkv, D, granules=gran, tag=tag[0], timeline=tl[l] if with_tl else None, scratch=scratch)

    def run_parts():
        for l in range(L):
            ops.gemv(x[None], w[l, nh * D:], norm_w=g, eps=1e-5, out=y)

    def run_attn():
        for l in range(L):
            ops.decode_attn_step(q_row, kc[l], vt[l], table, pos, nh, nkv, D, scratch=scratch)

    def timed(f, reps=5):
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / L * 1e3)
        return sorted(ts)[len(ts) // 2]

    out = {"pos": pos, "layers": L,
           "us_kv_attn_launch": round(timed(lambda: run(False)), 2),
           "us_kv_projection_alone": round(timed(run_parts), 2),
           "us_attention_alone": round(timed(run_attn), 2)}
    tl.zero_(); tl[:, 0] = big; tl[:, 6] = big
    run(True); torch.cuda.synchronize()
    t = tl.cpu().double()
    base = torch.minimum(t[:, 0], t[:, 6])
    names = ["attn0", "partial", "arrivals", "merged", "granules", "stored", "proj0", "proj_end", "last_merger", "last_chunk"]
    rel = (t - base[:, None]) / 100.0                     # 100 MHz ticks -> us
    out["timeline_us_median_over_layers"] = {n: round(float(rel[:, i].median()), 2) for i, n in enumerate(names)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
