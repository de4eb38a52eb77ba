"""Headline configurations at FULL depth against the oracle.

BASELINE config 2: LLaVA-1.5-7B geometry, 32 decoder + 23 executed CLIP layers, one 336 px image + 512-token prompt = 1087
positions, bf16; BASELINE config 3's model: LLaVA-1.5-13B geometry (40 layers), same request.  Weights are drawn once on the GPU
(HF init, std 0.02), rounded to bf16 and copied to the host, so the engine and the oracle hold bit-identical parameters.

Three host passes of oracle/llava_oracle.py per model over the same inputs:
  * fp32                     — the parity reference (the reference's CPU path is fp32: model_worker.py:139-141 keeps fp32 on cpu)
  * bf16, HF rounding points — what the REFERENCE itself computes in bf16 (every Linear / norm / activation output rounded, softmax
                               probabilities rounded before P·V, residual adds in bf16): the yardstick for bf16 noise
and one engine pass.  What is asserted, and why these numbers (measured on MI355X, 7B: image features engine 1.29e-2 / reference-bf16
1.31e-2 of max|ref|; last-position logits engine 2.7e-2 / reference-bf16 3.4e-2; 14 of 16 greedy ids identical, the other two 0.03 and
0.09 below the oracle's maximum logit of ~5.7):
  * bf16 engine vs fp32 oracle, last-position logits and image features: no further from fp32 than 1.25x the reference's OWN bf16 pass
    + 1 bf16 ulp (max-abs and rms), and never above 0.15 of max|ref|.  32 (40) layers of bf16 storage rounding put ANY bf16
    implementation ~3e-2 (13B: 6-9e-2: engine 6.4e-2, reference-bf16 8.9e-2) of max|logit| away from fp32 — that is the noise floor of the dtype, not of the kernels; the engine rounds less often than HF
    (SiLU·mul, residual adds and the softmax normalisation stay in fp32 inside the fused epilogues), so it lands closer to fp32.
  * greedy ids: the first 16 tokens, teacher-forced through the bf16 oracle: every id the engine picks must be the oracle's argmax or
    within the combined bf16 noise (engine + reference error of the prefill logits) of it, and at least half must be identical
    (measured 14-15 of 16 at 7B, 11-14 at 13B: near-ties — several at a gap of exactly 0 in bf16 — between random-init logits flip
    under ANY rounding change; a wrong kernel does not land within a few % of the maximum logit sixteen times in a row)
  * fp32 engine (exact-fp32 GEMM / attention verification mode) on an 8-layer cut at T = 1087: logits within 1e-3 of the fp32 oracle
    (north_star's tolerance), greedy ids identical.
Each run also writes its measured errors to gpurun_out/full_depth_<model>.json."""
import json
import os
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:  # noqa: BLE001
        pass
    return n


def _mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


def _build(cfg, dev, dtype, host=None):
    """Product model with synthetic HF-init weights.  host=None: draw on the GPU, return the bf16 host copy; else load `host`."""
    from llava_mi355x.model import LlavaLlamaForCausalLM
    from synthetic import build as harness, recipes as synth
    lc, vc = harness.hf_configs(cfg)
    model = LlavaLlamaForCausalLM(lc, vc, dtype=dtype, device=dev, max_position=2048)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    out = {}
    for name, shp in synth.tensor_shapes(cfg).items():
        if host is None:
            t = harness._device_tensor(cfg, name, shp, gen, dev).to(torch.bfloat16)
            out[name] = t.cpu()
        else:
            t = host[name]
        model.load_tensor(name, t)
    model.finalize_weights()
    model.get_vision_tower().is_loaded = True
    return model, (out if host is None else host)


def _layer_weights(w, i, dtype, cache=None):
    """decoder layer i's tensors in `dtype`; `cache` (a dict) keeps the upcast copies when the host has the memory for all of them"""
    if cache is not None and i in cache:
        return cache[i]
    p = f"model.layers.{i}."
    wl = {k: v.to(dtype) for k, v in w.items() if k.startswith(p)}
    if cache is not None:
        cache[i] = wl
    return wl


def _oracle_prefill(O, cfg, w, embeds, dtype, n_layers, want_past, all_rows=False, cache=None):
    """LlamaModel.forward + lm_head on the last row — or, all_rows, on every row, as transformers 4.31's LlamaForCausalLM does (llava_llama.py:88-99) —
    (oracle/llava_oracle.py: llama_forward), one decoder layer at a time so that the fp32 pass never needs more than one layer of upcast weights."""
    B, T, _ = embeds.shape
    pos = torch.arange(T)[None]
    cos, sin = O.rope_cos_sin(cfg, pos, dtype)
    qi = torch.arange(T)[:, None]; ki = torch.arange(T)[None, :]
    bias = torch.zeros((B, 1, T, T), dtype=dtype).masked_fill(~(ki <= qi)[None, None], torch.finfo(dtype).min)
    h = embeds.to(dtype)
    past = []
    for i in range(n_layers):
        wl = _layer_weights(w, i, dtype, cache)
        h, kv = O.decoder_layer(wl, cfg, i, h, cos, sin, None, bias)
        if want_past:
            past.append(kv)
    h = O.rms_norm(h if all_rows else h[:, -1:], w["model.norm.weight"].to(dtype), cfg.rms_norm_eps)
    lg = F.linear(h, w["lm_head.weight"].to(dtype))[0].float()
    return (lg if all_rows else lg[0]), past


def _front(O, cfg, w, ids, pix, dtype):
    """encode_images + splice (llava_arch.py:94-240) in `dtype`: (image features [576, H], inputs_embeds [1, T, H])."""
    small = {k: v.to(dtype) for k, v in w.items() if k.startswith("vision.") or k.startswith("mm_projector.") or k == "model.embed_tokens.weight"}
    feats = O.encode_images(small, cfg, pix.to(dtype))
    _, _, _, _, embeds, _ = O.prepare_inputs_labels_for_multimodal(small, cfg, ids, None, None, None, None, pix.to(dtype))
    return feats[0].float(), embeds


@pytest.mark.parametrize("name", ["llava15_7b", "llava15_13b"])
def test_full_depth_vs_oracle(cuda, name):
    import ctypes
    from llava_mi355x import _C
    from llava_mi355x.model import LmxKVCache
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    cfg = synth.CONFIGS[name]
    need_gb = sum(int(torch.tensor(s).prod()) for s in synth.tensor_shapes(cfg).values()) * 2 / 1e9 * 1.25 + 12
    if _mem_available_gb() < need_gb:
        pytest.skip(f"host has {_mem_available_gb():.0f} GB available, the {name} oracle needs ~{need_gb:.0f} GB")
    torch.set_num_threads(_usable_cores())
    report = {"model": name, "host_cores": torch.get_num_threads()}

    t0 = time.time()
    model, w = _build(cfg, cuda, torch.bfloat16)
    report["build_s"] = round(time.time() - t0, 1)
    ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1))
    V, L = cfg.vocab_size, cfg.num_hidden_layers

    # ---- engine: image features, last-position logits of the prefill, 16 greedy ids -------------------------------------------------
    pix_d = pix.to(cuda, torch.bfloat16); ids_d = ids.to(cuda)
    feats_e = model.encode_images(pix_d)[0].float().cpu()
    _, _, _, _, embeds_e, _ = model.prepare_inputs_labels_for_multimodal(ids_d, None, None, None, None, pix_d)
    assert embeds_e.shape[1] == 1087
    cache = LmxKVCache(model, 1)
    lg = torch.empty((1, V), dtype=torch.bfloat16, device=cuda)
    _C.check(_C.lib.lmx_prefill(model._h, cache.seqs[0], _C.ptr(embeds_e[0]), embeds_e.shape[1], 0, _C.ptr(lg), 0, 1, _C.stream_handle()))
    torch.cuda.synchronize()
    logits_e = lg[0].float().cpu()
    cache.close()
    N_TOK = 16
    gen = model.generate(inputs=ids_d, images=pix_d, do_sample=False, max_new_tokens=N_TOK, eos_token_id=-1)
    ids_e = gen[0, ids.shape[1]:].tolist()
    del model
    torch.cuda.empty_cache()

    with torch.no_grad():
        # ---- oracle, fp32 -----------------------------------------------------------------------------------------------------------
        t0 = time.time()
        feats_32, emb_32 = _front(O, cfg, w, ids, pix, torch.float32)
        logits_32, _ = _oracle_prefill(O, cfg, w, emb_32, torch.float32, L, False)
        report["oracle_fp32_s"] = round(time.time() - t0, 1)
        # ---- oracle, bf16 with HF's rounding points (keeps its KV cache for the teacher-forced steps) -----------------------------------
        t0 = time.time()
        feats_hf, emb_hf = _front(O, cfg, w, ids, pix, torch.bfloat16)
        logits_hf, past = _oracle_prefill(O, cfg, w, emb_hf, torch.bfloat16, L, True)
        report["oracle_bf16_s"] = round(time.time() - t0, 1)

        def errs(e, hf, ref):
            s = ref.abs().max().item()
            return {"scale": s, "engine": (e - ref).abs().max().item() / s, "hf_bf16": (hf - ref).abs().max().item() / s,
                    "engine_rms": ((e - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                    "hf_bf16_rms": ((hf - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()}

        report["image_features"] = errs(feats_e, feats_hf, feats_32)
        report["logits"] = errs(logits_e, logits_hf, logits_32)
        report["argmax"] = {"engine": int(logits_e.argmax()), "hf_bf16": int(logits_hf.argmax()), "fp32": int(logits_32.argmax())}

        # ---- greedy ids, teacher-forced on the engine's ids through the bf16 oracle ---------------------------------------------------
        noise = (report["logits"]["engine"] + report["logits"]["hf_bf16"]) * report["logits"]["scale"]
        steps = []
        cur = logits_hf
        for t in range(N_TOK):
            gap = (cur.max() - cur[ids_e[t]]).item()
            steps.append({"engine_id": ids_e[t], "oracle_id": int(cur.argmax()), "gap": gap})
            if t + 1 < N_TOK:
                emb = w["model.embed_tokens.weight"][torch.tensor([[ids_e[t]]])]
                nxt, past = O.llama_forward(w, cfg, emb, past=past, last_only=True)
                cur = nxt[0, 0].float()
        report["greedy"] = {"steps": steps, "identical": sum(s["engine_id"] == s["oracle_id"] for s in steps), "noise_abs": noise}
        del past

    from synthetic.treehash import csrc_sha16
    report["csrc_sha16"] = csrc_sha16()                       # the kernel sources these figures were measured on (bench.py: parity.stale)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"full_depth_{name}.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))

    ulp = 2.0 ** -8
    for key in ("image_features", "logits"):
        r = report[key]
        assert r["engine"] <= 0.15, f"{name} {key}: bf16 engine vs fp32 oracle {r['engine']:.3e} of max|ref|"
        assert r["engine"] <= 1.25 * r["hf_bf16"] + ulp, f"{name} {key}: engine {r['engine']:.3e} is further from fp32 than the reference's own bf16 pass {r['hf_bf16']:.3e}"
        assert r["engine_rms"] <= 1.25 * r["hf_bf16_rms"] + 1e-3, f"{name} {key}: rms {r['engine_rms']:.3e} vs {r['hf_bf16_rms']:.3e}"
    for t, s in enumerate(steps):
        assert s["gap"] <= noise + 1e-6, f"{name} greedy step {t}: engine id {s['engine_id']} is {s['gap']:.4f} below the bf16 oracle's maximum (noise {noise:.4f})"
    assert report["greedy"]["identical"] >= N_TOK // 2, f"{name}: only {report['greedy']['identical']} of {N_TOK} greedy ids equal the bf16 oracle's"


def test_fp32_engine_8_layers_T1087(cuda):
    """north_star's 'logits within 1e-3 of reference' at the headline prompt shape: the fp32 verification engine (same orchestration,
    splice, RoPE / KV / norm kernels; exact-fp32 GEMM and attention) on the first 8 decoder layers of the 7B geometry, T = 1087."""
    from llava_mi355x import _C
    from llava_mi355x.model import LmxKVCache
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    cfg = synth.with_layers(synth.CONFIGS["llava15_7b"], 8)
    torch.set_num_threads(_usable_cores())
    model, w = _build(cfg, cuda, torch.bfloat16)           # draw once in bf16-representable values ...
    del model
    torch.cuda.empty_cache()
    model, _ = _build(cfg, cuda, torch.float32, host=w)    # ... and hand the SAME values to the fp32 engine and the fp32 oracle
    ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1))
    out = model.generate(inputs=ids.to(cuda), images=pix.to(cuda), do_sample=False, max_new_tokens=4, eos_token_id=-1)
    _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids.to(cuda), None, None, None, None, pix.to(cuda))
    assert embeds.shape[1] == 1087
    cache = LmxKVCache(model, 1)
    lg = torch.empty((1, cfg.vocab_size), dtype=torch.float32, device=cuda)
    _C.check(_C.lib.lmx_prefill(model._h, cache.seqs[0], _C.ptr(embeds[0]), 1087, 0, _C.ptr(lg), 0, 1, _C.stream_handle()))
    torch.cuda.synchronize()
    cache.close()
    with torch.no_grad():
        w32 = {k: v.float() for k, v in w.items()}
        ref_logits, _, ref_emb, _ = O.llava_forward(w32, cfg, ids, pix, last_only=True)
        ref_tok = O.greedy_generate(w32, cfg, ids, pix, 4)
    err_emb = (embeds[0].cpu() - ref_emb[0]).abs().max().item()
    err = (lg[0].cpu() - ref_logits[0, 0]).abs().max().item()
    print(json.dumps({"fp32_8layer_T1087": {"logits_max_abs_err": err, "embeds_max_abs_err": err_emb, "max_abs_logit": ref_logits.abs().max().item()}}))
    assert err_emb <= 1e-3, f"spliced inputs_embeds max-abs-err {err_emb:.3e}"
    assert err <= 1e-3, f"fp32 engine logits max-abs-err {err:.3e}"
    assert out[0, ids.shape[1]:].tolist() == ref_tok


def _oracle_decode_step(O, cfg, w, emb, past, dtype, n_layers, cache=None):
    """One cached decode step of the oracle (decoder_layer with past (k, v)), one layer of upcast weights at a time."""
    S = past[0][0].shape[2]
    cos, sin = O.rope_cos_sin(cfg, torch.tensor([[S]]), dtype)
    h = emb.to(dtype)
    new_past = []
    for i in range(n_layers):
        wl = _layer_weights(w, i, dtype, cache)
        h, kv = O.decoder_layer(wl, cfg, i, h, cos, sin, past[i], None)
        new_past.append(kv)
    h = O.rms_norm(h, w["model.norm.weight"].to(dtype), cfg.rms_norm_eps)
    return F.linear(h, w["lm_head.weight"].to(dtype))[0, 0].float(), new_past


def test_fp32_engine_full_depth_T1087(cuda):
    """north_star's literal claim at its literal config (widened per VERDICT r4 item 7): LLaVA-1.5-7B geometry at FULL depth (32 decoder +
    23 executed CLIP layers), one 336 px image + 512-token prompt = 1087 positions, the fp32 verification engine (27 GB of fp32 weights in HBM) against the
    layer-streamed fp32 oracle: the logits of EVERY position (what transformers 4.31's forward returns, llava_llama.py:88-99) within 1e-3 absolute, and 32
    greedy ids identical (oracle teacher-forced on its own ids through its KV cache)."""
    from llava_mi355x import _C
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["llava15_7b"]
    if _mem_available_gb() < 40:
        pytest.skip(f"host has {_mem_available_gb():.0f} GB available")
    torch.set_num_threads(_usable_cores())
    L, N_TOK = cfg.num_hidden_layers, 32
    model, w = _build(cfg, cuda, torch.bfloat16)           # bf16-representable values, shared bit for bit by engine and oracle
    del model
    torch.cuda.empty_cache()
    model, _ = _build(cfg, cuda, torch.float32, host=w)
    ids = torch.from_numpy(synth.make_prompt(cfg, 512, image_positions=(35,), seed=2))[None]
    pix = torch.from_numpy(synth.make_pixels(cfg, 1, seed=1))
    t0 = time.time()
    out = model.generate(inputs=ids.to(cuda), images=pix.to(cuda), do_sample=False, max_new_tokens=N_TOK, eos_token_id=-1)
    ids_e = out[0, ids.shape[1]:].tolist()
    _, _, _, _, embeds, _ = model.prepare_inputs_labels_for_multimodal(ids.to(cuda), None, None, None, None, pix.to(cuda))
    assert embeds.shape[1] == 1087
    fwd = model.forward(input_ids=ids.to(cuda), images=pix.to(cuda), use_cache=False)      # logits of all 1087 positions, fp32
    torch.cuda.synchronize()
    engine_s = time.time() - t0
    assert fwd.logits.shape == (1, 1087, cfg.vocab_size)
    logits_e = fwd.logits[0].float().cpu()
    emb_e = embeds[0].cpu()
    del model, fwd
    torch.cuda.empty_cache()
    cache = {} if _mem_available_gb() > 70 else None      # 26 GB of upcast decoder weights kept across the 32 oracle steps when the host has the room
    with torch.no_grad():
        t0 = time.time()
        _, emb_32 = _front(O, cfg, w, ids, pix, torch.float32)
        logits_32, past = _oracle_prefill(O, cfg, w, emb_32, torch.float32, L, True, all_rows=True, cache=cache)
        cur = logits_32[-1]
        ids_o = []
        for t in range(N_TOK):
            ids_o.append(int(cur.argmax()))
            if t + 1 < N_TOK:
                emb = w["model.embed_tokens.weight"][torch.tensor([[ids_o[-1]]])]
                cur, past = _oracle_decode_step(O, cfg, w, emb, past, torch.float32, L, cache=cache)
        oracle_s = time.time() - t0
    err_emb = (emb_e - emb_32[0]).abs().max().item()
    err_rows = (logits_e - logits_32).abs().amax(dim=1)
    err, err_last = err_rows.max().item(), err_rows[-1].item()
    rep = {"fp32_full_depth_T1087": {"layers": L, "positions_compared": int(err_rows.numel()), "logits_max_abs_err_all_positions": err, "logits_max_abs_err_last_position": err_last,
                                     "embeds_max_abs_err": err_emb, "max_abs_logit": logits_32.abs().max().item(), "greedy_ids_compared": N_TOK,
                                     "greedy_ids_identical": sum(int(a == b) for a, b in zip(ids_e, ids_o)),
                                     "engine_ids": ids_e, "oracle_ids": ids_o, "engine_s": round(engine_s, 1), "oracle_s": round(oracle_s, 1)}}
    from synthetic.treehash import csrc_sha16
    rep["csrc_sha16"] = csrc_sha16()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "full_depth_fp32_llava15_7b.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    assert err_emb <= 1e-3, f"spliced inputs_embeds max-abs-err {err_emb:.3e}"
    assert err <= 1e-3, f"fp32 engine logits max-abs-err {err:.3e} over all 1087 positions at 32 layers"
    assert ids_e == ids_o
