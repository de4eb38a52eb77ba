"""BASELINE config 4 harness (SURVEY §8 f-4 second half, §8d): a LLaVA-Plus model worker serving CONCURRENT requests that each walk the
tool loop of llava/serve/gradio_web_server_llava_plus.py:444-637 against stub `grounding_dino` / `sam` REST workers:

    client -> POST /worker_generate_stream (prompt + image)  -> streamed answer
           -> parse  "thoughts🤔" ... "actions🚀" [ {API_name, API_params} ] "value👉" ...            (:498-516)
           -> POST <tool worker>/worker_generate {image, box_threshold, text_threshold, **API_params}   (:520-560)
           -> clean the tool response (round boxes / logits, drop masks_rle / size)                     (:565-596)
           -> re-prompt "<api> model outputs: {...}\n\nPlease summarize the model outputs and answer my first question: ..." (:600-610)
           -> POST /worker_generate_stream again -> final answer                                        (:612-637)

The stub tool workers answer with the JSON the reference's workers produce (serve/grounding_dino_worker.py:231-236:
{boxes, logits, phrases, size}; serve/sam_worker.py:252-255: {masks_rle, boxes}).  The model is the scripted model of
synthetic/scripted.py (real geometry and kernels, weights arranged so that greedy decoding recites a tool call, then a summary), served
through tools/worker_reenactment.py with the continuous-batching scheduler on.  Everything from the HTTP request to the streamed text is
the product path.

    python tools/config4_harness.py --model llava_plus_v0_13b --requests 32 [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tools/config4_harness.py ...     (TP=8: rank 0
        serves, ranks 1-7 follow its command log; add --shared-gpu on a one-GPU box, where the ranks meet through HIP IPC instead of RCCL)"""
import argparse
import base64
import copy
import io
import json
import os
import re
import socket
import sys
import threading
import time
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "llava-plus-codebase_amd"), ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

R = partial(round, ndigits=2)                       # gradio_web_server_llava_plus.py:36
SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. "
          "The assistant gives helpful, detailed, and polite answers to the human's questions.")     # conv_llava_v1 (conversation.py:599-609)
SEP, SEP2 = " ", "</s>"


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def b64_image(seed, size=(96, 64)):
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(seed)
    img = Image.fromarray(rng.randint(0, 255, size=(size[1], size[0], 3), dtype=np.uint8), "RGB")
    buf = io.BytesIO(); img.save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode()


def get_prompt(messages):
    """Conversation.get_prompt for SeparatorStyle.TWO (conversation.py:97-106): system + sep, then `role: message` + sep / sep2."""
    ret = SYSTEM + SEP
    for i, (role, message) in enumerate(messages):
        ret += (role + ": " + message + (SEP, SEP2)[i % 2]) if message else (role + ":")
    return ret


def make_tool_app(kind, calls, latency_s=0.005):
    """Stub tool worker: /worker_generate with the reference worker's response schema."""
    from fastapi import FastAPI, Request
    app = FastAPI()

    @app.post("/worker_generate")
    async def worker_generate(request: Request):
        params = await request.json()
        calls.append((kind, sorted(params.keys())))
        time.sleep(latency_s)
        if kind == "grounding_dino":        # serve/grounding_dino_worker.py:199-236
            assert "caption" in params and "image" in params and "box_threshold" in params and "text_threshold" in params
            return {"boxes": [[0.123456, 0.2, 0.654321, 0.7]], "logits": [0.876543], "phrases": [params["caption"].strip(" .")], "size": [64, 96]}
        assert "image" in params            # serve/sam_worker.py:195-255
        return {"masks_rle": [{"size": [64, 96], "counts": "0000"}], "boxes": params.get("boxes", [])}

    return app


def stream_answer(worker_addr, pload, timeout=120):
    """requests.post(... stream=True) + iter_lines(delimiter=b'\\0')  (gradio_web_server_llava_plus.py:462-484). Returns (text, first-chunk time)."""
    import requests
    t0 = time.perf_counter(); t_first = None
    response = requests.post(worker_addr + "/worker_generate_stream", headers={"User-Agent": "LLaVA Client"}, json=pload, stream=True, timeout=timeout)
    output = ""
    for chunk in response.iter_lines(decode_unicode=False, delimiter=b"\0"):
        if chunk:
            data = json.loads(chunk.decode())
            if t_first is None:
                t_first = time.perf_counter() - t0
            if data["error_code"] != 0:
                raise RuntimeError(data["text"] + f" (error_code: {data['error_code']})")
            output = data["text"][len(pload["prompt"]):].strip()
    return output, t_first, time.perf_counter() - t0


def tool_loop(worker_addr, tool_addrs, question, image_b64, max_new_tokens=256):
    """One chat turn with tool augmentation, as the Gradio server drives it."""
    import requests
    roles = ("USER", "ASSISTANT")
    messages = [[roles[0], "<image>\n" + question], [roles[1], None]]
    prompt = get_prompt(messages)
    pload = {"model": "llava-scripted", "prompt": prompt, "temperature": 0.0, "top_p": 1.0, "max_new_tokens": min(int(max_new_tokens), 1536),
             "stop": SEP2, "images": [image_b64]}
    out1, ttft1, lat1 = stream_answer(worker_addr, pload)
    messages[-1][1] = out1
    rec = {"first_answer": out1, "ttft_s": ttft1, "round1_s": lat1, "tool": None}
    # parse the output (:498-516)
    tool_cfg = None
    matches = re.findall(r'"thoughts🤔"(.*)"actions🚀"(.*)"value👉"(.*)', out1, re.DOTALL)
    if matches:
        try:
            tool_cfg = json.loads(matches[0][1].strip())
        except Exception:  # noqa: BLE001
            tool_cfg = json.loads(matches[0][1].strip().replace("'", '"'))
    if tool_cfg:
        assert len(tool_cfg) == 1, "Only one tool is supported for now"
        api_name = tool_cfg[0]["API_name"]
        tool_cfg[0]["API_params"].pop("image", None)
        api_paras = {"image": image_b64, "box_threshold": 0.3, "text_threshold": 0.25, **tool_cfg[0]["API_params"]}
        t0 = time.perf_counter()
        tool_response = requests.post(tool_addrs[api_name] + "/worker_generate", headers={"User-Agent": "LLaVA Client"}, json=api_paras).json()
        rec["tool"] = api_name; rec["tool_s"] = time.perf_counter() - t0
        rec["tool_response_raw"] = copy.deepcopy(tool_response)
        if "boxes" in tool_response:
            tool_response["boxes"] = [[R(_b) for _b in bb] for bb in tool_response["boxes"]]
        if "logits" in tool_response:
            tool_response["logits"] = [R(_l) for _l in tool_response["logits"]]
        masks_rle = tool_response.pop("masks_rle", None)
        tool_response.pop("size", None)
        if len(tool_response) == 0:
            tool_response["message"] = f"The {api_name} has processed the image."
        rec["mask_rle"] = masks_rle[0] if masks_rle else None
        new_response = f"{api_name} model outputs: {tool_response}\n\n"
        first_question = messages[-2][1].replace("<image>", "").strip()
        messages.append([roles[0], new_response + "Please summarize the model outputs and answer my first question: {}".format(first_question)])
        messages.append([roles[1], None])
        prompt2 = get_prompt(messages)
        pload2 = dict(pload, prompt=prompt2)
        pload2.pop("top_p")
        out2, ttft2, lat2 = stream_answer(worker_addr, pload2)
        rec.update({"final_answer": out2, "ttft2_s": ttft2, "round2_s": lat2, "prompt2_tail": prompt2[-400:]})
    rec["total_s"] = rec["round1_s"] + rec.get("tool_s", 0.0) + rec.get("round2_s", 0.0)
    return rec


def tp_setup(shared_gpu: bool):
    """One process per GPU (torchrun env: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, device, group) — group is the
    gloo group of the command channel.  shared_gpu: every rank on cuda:0 (a one-GPU box: RCCL refuses that, all-reduces go peer-to-peer)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, "cuda:0", None
    dev = 0 if shared_gpu else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(dev)
    if shared_gpu:
        os.environ["LMX_TP_P2P_ALL"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    return rank, world, f"cuda:{dev}", dist.new_group(backend="gloo")


def run(cfg_name, n_requests, batch, dtype_name="bf16", concurrency=None, sam_every=4, max_new_tokens=256, shared_gpu=False, packed=True, reuse=True, prefill_thread=2):
    import torch
    from synthetic import recipes as synth, scripted
    import worker_reenactment as wr
    cfg = synth.CONFIGS[cfg_name]
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype_name]
    rank, world, device, group = tp_setup(shared_gpu)
    tok, model, proc, _ = scripted.build_scripted(cfg, dtype=dt, device=device, tp_rank=rank, tp_world=world)
    channel = None
    if world > 1:
        # tensor parallel: rank 0 is the worker the clients talk to, the other ranks replay its command log (llava_mi355x/tp_serving.py)
        from llava_mi355x import tp_serving
        model.init_tensor_parallel(rccl=not shared_gpu, p2p=True)
        channel = tp_serving.CommandChannel(group)
        if rank != 0:
            stats = tp_serving.serve_follower(model, channel, capacity=max(batch, 1))
            return {"rank": rank, "follower": stats}
    if batch > 1 or world > 1:
        model.enable_batching(capacity=max(batch, 1), channel=channel, packed_prefill=packed, prefill_thread=prefill_thread)
    if reuse:
        # the tool loop's second generate re-sends the image and the whole first exchange (gradio_web_server_llava_plus.py:600-637): image features by pixel
        # content, KV rows of the finished first turn taken over by the second (llava_mi355x/reuse.py); one entry per conversation in flight
        model.enable_reuse(images=max(64, 2 * n_requests), prefixes=max(32, n_requests + 1))
    calls = []
    ports = {"worker": free_port(), "grounding_dino": free_port(), "sam": free_port()}
    servers = [wr.serve_in_thread(wr.make_worker_app(tok, model, proc, limit_model_concurrency=concurrency or max(5, n_requests)), ports["worker"]),
               wr.serve_in_thread(make_tool_app("grounding_dino", calls), ports["grounding_dino"]),
               wr.serve_in_thread(make_tool_app("sam", calls), ports["sam"])]
    addr = {k: f"http://127.0.0.1:{v}" for k, v in ports.items()}
    recs, errors = [None] * n_requests, []

    def client(i):
        try:
            q = "Please segment the object." if sam_every and i % sam_every == sam_every - 1 else "Where is the object?"
            recs[i] = tool_loop(addr["worker"], addr, q, b64_image(100 + i), max_new_tokens)
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    tool_loop(addr["worker"], addr, "Where is the object?", b64_image(7), max_new_tokens)        # warm-up (allocations, first launches)
    calls.clear()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=client, args=(i,)) for i in range(n_requests)]
    for t in ths: t.start()
    for t in ths: t.join()
    wall = time.perf_counter() - t0
    sched = None
    if batch > 1 or world > 1:
        bt = model._batcher
        sched = {"prefill_thread": bool(bt.prefill_thread), "prepare_on_request_thread": bool(bt.prepare_on_request_thread), "decode_steps": bt.steps, "member_steps": bt.member_steps, "max_live": bt.max_live,
                 "packed_prefills": bt.prefill_batches, "requests_prefilled": bt.prefilled}
        model.disable_batching()                       # under TP this also tells the followers to stop
    for s, _ in servers:
        s.should_exit = True
    ok = [r for r in recs if r]
    n_tok = sum(len(scripted._pieces(r["first_answer"])) + len(scripted._pieces(r.get("final_answer", ""))) for r in ok)
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else None
    dist = lambda xs: {"min": round(min(xs) * 1e3, 1), "p50": round(sorted(xs)[len(xs) // 2] * 1e3, 1), "p90": round(sorted(xs)[(len(xs) * 9) // 10] * 1e3, 1),
                       "max": round(max(xs) * 1e3, 1)} if xs else None
    return {"workload": f"config4: {cfg_name} scripted model, {n_requests} concurrent tool-loop requests (generate -> parse actions -> stub tool worker "
                        f"-> re-prompt -> generate), decode batch capacity {batch}, TP={world}" + (" (ranks share one GPU, peer-to-peer all-reduce)" if world > 1 and shared_gpu else ""),
            "tp_world": world, "rccl_ranks": model.tp_comm_ranks(), "p2p_active": bool(getattr(model, "p2p_active", False)), "requests": n_requests, "completed": len(ok), "errors": errors,
            "wall_s": wall, "generated_tokens_per_s": n_tok / wall, "tool_calls": {k: sum(1 for c in calls if c[0] == k) for k in ("grounding_dino", "sam")},
            "median_ttft_s": med([r["ttft_s"] for r in ok]), "median_total_s": med([r["total_s"] for r in ok]),
            "median_round2_ttft_s": med([r["ttft2_s"] for r in ok if "ttft2_s" in r]),
            "ttft_ms": dist([r["ttft_s"] for r in ok]), "round2_ttft_ms": dist([r["ttft2_s"] for r in ok if "ttft2_s" in r]),
            "round1_ms": dist([r["round1_s"] for r in ok]), "round2_ms": dist([r["round2_s"] for r in ok if "round2_s" in r]), "scheduler": sched, "reuse": model.reuse_stats() if reuse else None, "records": recs, "expected": {"tool": scripted.TOOL_CALL, "sam": scripted.SAM_CALL, "summary": scripted.SUMMARY}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llava_plus_v0_13b")
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--packed", type=int, default=1, help="1: the scheduler prefills waiting requests together (default); 0: one prefill per request thread")
    ap.add_argument("--shared-gpu", action="store_true", help="TP ranks all on cuda:0 (one-GPU box; no RCCL, peer-to-peer all-reduce)")
    ap.add_argument("--prefill-thread", type=int, default=2, help="2 (default): packed prefills on their own thread and stream beside the decode steps, rank-local half on the request threads; 1: all of it on the prefill thread; 0: between two decode steps")
    ap.add_argument("--reuse", type=int, default=1, help="1 (default): image-feature cache + KV prefix reuse between the two turns of a conversation; 0: every turn from scratch")
    a = ap.parse_args()
    res = run(a.model, a.requests, a.batch, a.dtype, shared_gpu=a.shared_gpu, packed=bool(a.packed), reuse=bool(a.reuse), prefill_thread=int(a.prefill_thread))
    if "records" in res:
        recs = res.pop("records"); exp = res.pop("expected")
        res["sample_final_answer"] = next((r.get("final_answer") for r in recs if r), None)
        res["answers_as_scripted"] = sum(1 for r in recs if r and r.get("final_answer") == exp["summary"])
    print(json.dumps(res), flush=True)
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier(); dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
