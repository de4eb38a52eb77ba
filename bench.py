#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X LLaVA forward path (BASELINE.json metric).

Workload (BASELINE.json configs[1]; name in `config.workload`): LLaVA-1.5-7B geometry, bf16, one synthetic 336x336
image + 512-token prompt (one <image> marker -> 1087 positions), greedy 128-token generation.  Random-init weights
(HF init, std 0.02 — no checkpoints exist offline), synthetic inputs, already resident in HBM when timing starts.

One "step" = one whole request through the hot path: encode_images (CLIP ViT-L/14-336, 23 layers, + mlp2x_gelu
projector) -> splice -> decoder prefill (1087 positions) -> 127 decode steps with the KV cache (+ the prefill's pick
= 128 generated tokens).  `value` = generated tokens / wall second over the timed steps, whole job.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU,
                                                        the decoder runs tensor-parallel over RCCL: scaling = strong)
Extra objects on the JSON line (measured in the same process after the timed region, HIP events on the launch stream):
  roofline          dominant kernel of the step by time = decode weight-streaming GEMV family (HBM-bound)
  roofline_prefill  dominant prefill kernel = MFMA GEMM family (MFMA-bound; north_star's 40 % target)
  cpu_baseline      the CPU oracle (oracle/llava_oracle.py, torch CPU) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0         # MI355X HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llava15_7b")
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the serving_batch (continuous batching) measurement")
    ap.add_argument("--no-replicas", action="store_true", help="N>1: skip the independent-replicas (data-parallel serving) measurement")
    ap.add_argument("--cpu-layers", type=int, default=2, help="decoder layers in the bounded CPU sample")
    ap.add_argument("--gemm-variant", type=int, default=0)
    return ap.parse_args()


def flops_prefill(cfg, T, n_images):
    H, I, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    lin = 2.0 * T * L * (4 * H * H + 3 * H * I)
    attn = 2.0 * T * T * H * L                       # causal: QK^T + PV over the lower triangle
    head = 2.0 * H * cfg.vocab_size
    Dv, Fv, Tv, P = cfg.v_hidden_size, cfg.v_intermediate_size, cfg.num_patches + 1, cfg.num_patches
    n_hs = cfg.v_num_hidden_layers + 1
    v_run = cfg.mm_vision_select_layer % n_hs
    vis = v_run * (2.0 * Tv * (4 * Dv * Dv + 2 * Dv * Fv) + 4.0 * Tv * Tv * Dv) + 2.0 * P * 3 * cfg.v_patch_size ** 2 * Dv
    proj = 2.0 * P * (Dv * H + H * H)
    return dict(linear=lin, attention=attn, lm_head=head, vision=n_images * vis, projector=n_images * proj,
                total=lin + attn + head + n_images * (vis + proj))


def usable_cores() -> int:
    """Cores this process may really use: affinity mask, clipped by the cgroup CPU quota (containers often expose every
    host CPU in os.cpu_count() while being limited to a few)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:  # noqa: BLE001
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:  # noqa: BLE001
            pass
    return n


def cpu_baseline(cfg, T, new_tokens, n_layers):
    """Bounded sample of the same workload on the host cores through the CPU oracle (torch CPU), in fp32 and bf16:
    full CLIP tower + projector for 1 image, `n_layers` real-geometry decoder layers of prefill at T positions and of
    4 decode steps at ctx T; decoder time scaled by L / n_layers, lm_head timed once.  `value` is the faster dtype
    (torch's CPU bf16 GEMM is far slower than fp32 on hosts without AMX / AVX512-BF16 kernels)."""
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    torch.set_num_threads(usable_cores())
    small = synth.with_layers(cfg, n_layers)
    g = torch.Generator().manual_seed(0)
    w32 = {}
    for name, shp in synth.tensor_shapes(small).items():
        if name.endswith("norm.weight") or ("layer_norm" in name and name.endswith(".weight")) or name.endswith("pre_layrnorm.weight"):
            w32[name] = torch.ones(shp)
        elif name.endswith(".bias"):
            w32[name] = torch.zeros(shp)
        else:
            w32[name] = torch.randn(shp, generator=g) * 0.02
    pix32 = torch.from_numpy(synth.make_pixels(cfg, 1))
    emb32 = torch.randn((1, T, cfg.hidden_size), generator=g) * 0.02
    scale = cfg.num_hidden_layers / n_layers
    nd = 4
    per = {}
    for dname, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        if per and per["fp32"]["sample_wall_s"] > 15.0:
            break                                        # keep the whole bench within a few minutes on slow hosts
        w = {k: v.to(dt) for k, v in w32.items()}
        pix, emb = pix32.to(dt), emb32.to(dt)
        with torch.no_grad():
            t0 = time.perf_counter(); O.encode_images(w, small, pix); t_vis = time.perf_counter() - t0
            t0 = time.perf_counter(); _, past = O.llama_forward(w, small, emb, last_only=True); t_pre = time.perf_counter() - t0
            tok = emb[:, :1]
            t0 = time.perf_counter()
            for _ in range(nd):
                _, past = O.llama_forward(w, small, tok, past=past, last_only=True)
            t_dec = (time.perf_counter() - t0) / nd
            h = emb[:, -1:]
            t0 = time.perf_counter(); O.rms_norm(h, w["model.norm.weight"], cfg.rms_norm_eps) @ w["lm_head.weight"].t(); t_head = time.perf_counter() - t0
        prefill_s = t_vis + max(t_pre - t_head, 0.0) * scale + t_head
        decode_s = max(t_dec - t_head, 0.0) * scale + t_head
        step_s = prefill_s + (new_tokens - 1) * decode_s
        per[dname] = {"tokens_per_s": new_tokens / step_s, "prefill_ms": prefill_s * 1e3, "decode_tokens_per_s": 1.0 / decode_s,
                      "sample_wall_s": t_vis + t_pre + nd * t_dec + t_head}
    best = max(per, key=lambda k: per[k]["tokens_per_s"])
    return {"value": per[best]["tokens_per_s"], "unit": "generated tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "dtype": best, "prefill_ms": per[best]["prefill_ms"], "decode_tokens_per_s": per[best]["decode_tokens_per_s"], "by_dtype": per,
            "sample": f"full CLIP tower+projector (1 image) + {n_layers}/{cfg.num_hidden_layers} decoder layers at real 7B geometry: "
                      f"prefill T={T} and {nd} decode steps at ctx {T}; decoder time scaled x{scale:g}, lm_head timed once"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # LMX_BENCH_SHARE_GPU=1 (dry run of the N>1 code path on a one-GPU box): every rank uses GPU 0, gloo instead of RCCL for
    # the host-side rendezvous and the engine's all-reduces all go through the P2P kernel (set LMX_TP_P2P_ALL=1 as well)
    share = os.environ.get("LMX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from synthetic import build as harness, recipes as synth
    cfg = synth.CONFIGS[a.model]
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    t0 = time.time()
    model = harness.build_model(cfg, dtype=dtype, seed=0, device_rng=True, device=dev, tp_rank=rank, tp_world=world,
                                max_position=2048, gemm_variant=a.gemm_variant)
    model.init_tensor_parallel(rccl=not share)
    build_s = time.time() - t0

    ids = torch.from_numpy(synth.make_prompt(cfg, a.prompt_len, image_positions=(35,), seed=2))[None].to(dev)
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1)).to(dev, dtype)
    T = a.prompt_len - 1 + cfg.tokens_per_image
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier() if share else dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    def step():
        out = model.generate(inputs=ids, images=pix, do_sample=False, max_new_tokens=a.new_tokens, eos_token_id=-1, run_ahead=a.new_tokens)
        assert out.shape[1] == a.prompt_len + a.new_tokens
        return out

    outs = []
    for _ in range(a.warmup):
        outs.append(step())
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs.append(step())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    deterministic = all(torch.equal(outs[0], o) for o in outs[1:])     # same request -> same greedy ids every step
    ms_per_step = dt / a.steps * 1e3
    value = a.new_tokens * a.steps / dt

    # ---- phase split + per-kernel durations: one profiled replay of the same step (HIP events on the launch stream) ----
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    from llava_mi355x.model import LmxKVCache
    from llava_mi355x import _C
    prefill_ms, decode_ms = [], []
    for _ in range(max(2, a.steps)):
        barrier()
        e[0].record()
        _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
        cache = LmxKVCache(model, 1)
        _C.check(_C.lib.lmx_prefill(model._h, cache.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
        e[1].record()
        _C.check(_C.lib.lmx_decode(model._h, cache.seqs[0], -1, a.new_tokens - 1, None, 1, _C.stream_handle()))
        e[2].record()
        torch.cuda.synchronize()
        prefill_ms.append(e[0].elapsed_time(e[1])); decode_ms.append(e[1].elapsed_time(e[2]))
        cache.close()
    prefill_ms = sorted(prefill_ms)[len(prefill_ms) // 2]
    decode_ms = sorted(decode_ms)[len(decode_ms) // 2]

    model.profile(True)
    step()
    prof = model.profile_read()
    model.profile(False)

    fl = flops_prefill(cfg, T, 1)
    H, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size
    # decode GEMV family: algorithmic bytes = weight bytes streamed per launch (13.214 GB / token for 7B)
    es = 2
    gemv_bytes = {"decode.gemv.qkv": 3 * H * H * es / world, "decode.gemv.o": H * H * es / world, "decode.gemv.gate_up": 2 * H * I * es / world,
                  "decode.gemv.down": H * I * es / world, "decode.gemv.lm_head": V * H * es}
    gb = sum(gemv_bytes[k] * prof[k][1] for k in gemv_bytes if k in prof)
    gs = sum(prof[k][0] for k in gemv_bytes if k in prof) * 1e-3
    n_gemv = sum(prof[k][1] for k in gemv_bytes if k in prof)
    # HBM-side traffic per launch from the committed PMC passes (profiles/r01_pmc_traffic.json): bytes fetched+written by the
    # GEMV family per generated token divided by its launches; rocprofv3 --pmc cannot wrap this process (it segfaults here),
    # so the counters come from the cold-cache single-kernel driver at the same shapes.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pt = json.load(f)["gemv_kernel"]
        if world == 1 and a.model == "llava15_7b":
            per_layer = sum(v["hbm_bytes"] for v in pt.values())
            ratio = per_layer / sum(v["algorithmic_bytes"] for v in pt.values())
            traffic = ratio * gb / max(n_gemv, 1)
    except Exception:  # noqa: BLE001
        traffic = None
    roof = {"bound": "hbm", "kernel": "gemv_kernel<bf16,1,R> (decode linears incl. fused RMSNorm / SiLU·mul / residual)",
            "achieved": gb / gs / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gb / gs / 1e9 / PEAK_HBM_GBS,
            "launches": int(n_gemv), "avg_launch_us": gs / max(n_gemv, 1) * 1e6, "bytes_per_token": 4 * H * H * es * L / world + 3 * H * I * es * L / world + V * H * es,
            "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC, profiles/r01_pmc_traffic.json); algorithmic = %.0f" % (gb / max(n_gemv, 1)),
            "measured": "HIP events around every launch, profiled replay of the timed step (same process/stream)"}
    gemm_flops = {"prefill.gemm.qkv": 2.0 * T * 3 * H * H / world, "prefill.gemm.o": 2.0 * T * H * H / world,
                  "prefill.gemm.gate_up": 2.0 * T * 2 * H * I / world, "prefill.gemm.down": 2.0 * T * H * I / world}
    gf = sum(gemm_flops[k] * prof[k][1] for k in gemm_flops if k in prof)
    gt = sum(prof[k][0] for k in gemm_flops if k in prof) * 1e-3
    n_gemm = sum(prof[k][1] for k in gemm_flops if k in prof)
    roof_p = {"bound": "mfma", "kernel": "gemm_mfma_kernel<bf16,...> (decoder prefill linears)", "achieved": gf / gt / 1e12, "peak": PEAK_BF16_TFLOPS,
              "unit": "TFLOP/s", "frac": gf / gt / 1e12 / PEAK_BF16_TFLOPS, "launches": int(n_gemm), "avg_launch_us": gt / max(n_gemm, 1) * 1e6,
              "traffic": None, "prefill_total_tflop": fl["total"] / 1e12 / (1 if world == 1 else 1),
              "prefill_end_to_end_frac": fl["total"] / (prefill_ms * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS * world),
              "measured": "HIP events around every launch, profiled replay of the timed step (same process/stream)"}
    breakdown = {k: {"ms": round(v[0], 4), "n": int(v[1])} for k, v in sorted(prof.items())}
    # at TP=1 the all-reduce scope brackets nothing: its elapsed time is the pure cost of a HIP event pair on the stream, i.e. the
    # amount by which every per-launch figure above overstates the kernel's own duration (rocprofv3 sees the kernel alone)
    marker_us = prof["decode.allreduce"][0] / max(prof["decode.allreduce"][1], 1) * 1e3 if world == 1 and "decode.allreduce" in prof else None
    roof["event_pair_overhead_us"] = marker_us
    roof_p["event_pair_overhead_us"] = marker_us

    # ---- N > 1: the same N GPUs as N independent replicas (SURVEY §8e "data-parallel serving fallback": one full model per GPU, one
    #      request each, no collective) — reported next to the tensor-parallel `value`, never instead of it
    replicas = None
    if world > 1 and not a.no_replicas:
        import torch.distributed as dist
        del outs
        full = harness.build_model(cfg, dtype=dtype, seed=0, device_rng=True, device=dev, tp_rank=0, tp_world=1, max_position=2048)
        ids_r = torch.from_numpy(synth.make_prompt(cfg, a.prompt_len, image_positions=(35,), seed=2 + rank))[None].to(dev)

        def step_r():
            return full.generate(inputs=ids_r, images=pix, do_sample=False, max_new_tokens=a.new_tokens, eos_token_id=-1, run_ahead=a.new_tokens)

        for _ in range(max(1, a.warmup)):
            step_r()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step_r()
        barrier()
        dtr = time.perf_counter() - t0
        ttr = torch.tensor([dtr], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(ttr, op=dist.ReduceOp.MAX)
        dtr = float(ttr.item())
        replicas = {"what": f"{world} independent full-model replicas, one request each, no collective", "scaling": "weak",
                    "value": world * a.new_tokens * a.steps / dtr, "ms_per_step": dtr / a.steps * 1e3}
        del full
        torch.cuda.empty_cache()

    # ---- serving-side view (BASELINE configs 3/4 run many requests at once): the same request x B decoding together through
    #      lmx_decode_batch (continuous batching) — aggregate generated tokens/s of the decode phase, not part of `value`
    serving = None
    if world == 1 and not a.no_batch:
        from llava_mi355x.batching import DecodeBatch
        serving = {"what": "decode phase of B identical config-2 requests stepping together (lmx_decode_batch), context %d" % T, "by_batch": {}}
        _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, pix)
        sizes = [8, 16, 32]
        caches = []
        for _ in range(max(sizes)):
            c = LmxKVCache(model, 1)
            _C.check(_C.lib.lmx_prefill(model._h, c.seqs[0], _C.ptr(embeds[0]), embeds.shape[1], 0, None, 0, 1, _C.stream_handle()))
            caches.append(c)
        bt = DecodeBatch(model, max(sizes))
        for Bn in sizes:
            seqs = [c.seqs[0] for c in caches[:Bn]]
            bt.step(seqs, None, 2, True, want_ids=False)
            torch.cuda.synchronize()
            e[0].record(); bt.step(seqs, None, 16, True, want_ids=False); e[1].record()
            torch.cuda.synchronize()
            ms = e[0].elapsed_time(e[1]) / 16
            serving["by_batch"][str(Bn)] = {"ms_per_step": ms, "decode_tokens_per_s": Bn * 1e3 / ms}
        bt.close()
        for c in caches:
            c.close()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cfg, T, a.new_tokens, a.cpu_layers)
        except Exception as ex:  # noqa: BLE001
            cpu = {"error": repr(ex)}

    # Native libraries write to C stdio (RCCL prints a version banner on its first communicator), which is block-buffered on a pipe and
    # would otherwise come out at process exit, AFTER the JSON line.  Every rank flushes it now, then rank 0 prints the JSON line last.
    try:
        import ctypes as _ct
        _ct.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    barrier()
    if rank == 0:
        line = {"metric": "generated tokens/sec + prefill ms (336px img + 512-tok prompt), LLaVA-1.5-7B", "value": value,
                "unit": "generated tokens/s (whole request: image encode + prefill + decode)", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
                "dtype": a.dtype, "data": "synthetic (seeded image + ids, random-init HF-std weights)",
                "config": {"workload": f"{a.model}: 1x336x336 image + {a.prompt_len}-token prompt ({T} positions), greedy {a.new_tokens} new tokens, batch 1",
                           "parallelism": f"tp{world}", "kv_capacity": 2048,
                           "decode_allreduce": ("p2p-one-shot" if getattr(model, "p2p_active", False) else "rccl") if world > 1 else None},
                "prefill_ms": prefill_ms, "decode_tokens_per_s": (a.new_tokens - 1) / (decode_ms * 1e-3), "decode_ms_per_token": decode_ms / (a.new_tokens - 1),
                "roofline": roof, "roofline_prefill": roof_p, "cpu_baseline": cpu, "serving_batch": serving, "replicas": replicas, "kernel_breakdown_ms_per_step": breakdown,
                "model_build_s": build_s, "greedy_ids_identical_across_steps": bool(deterministic)}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
