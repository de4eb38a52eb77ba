"""The reference model worker's request path, re-enacted call for call on this build (the unmodified llava/serve/model_worker.py cannot
be imported on the GPU box: /root/reference is not there; tests/test_unmodified_worker_boundary.py imports it in the build container).

  generate_stream(...)   = ModelWorker.generate_stream                      llava/serve/model_worker.py:122-192
  make_worker_app(...)   = the FastAPI app: /worker_generate_stream (NUL-delimited JSON chunks behind a concurrency semaphore),
                           /worker_get_status                                 model_worker.py:221-244
  serve_in_thread(...)   = uvicorn.run on a daemon thread (the reference runs it in the main thread)
Used by tests/test_worker_flow_gpu.py, tests/test_tool_loop_gpu.py and tools/config4_harness.py."""
import asyncio
import json
import threading
import time


def generate_stream(tokenizer, model, image_processor, params, streamer_timeout=60):
    """Same steps as the worker's generate_stream; yields the JSON-serialisable chunks it would send."""
    import torch
    from threading import Thread
    from transformers import TextIteratorStreamer
    from llava_mi355x.constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
    from llava_mi355x.mm_utils import KeywordsStoppingCriteria, load_image_from_base64, process_images, tokenizer_image_token
    prompt = params["prompt"]; ori_prompt = prompt
    images = params.get("images")
    num_image_tokens = 0
    image_args = {}
    if images:
        if len(images) != prompt.count(DEFAULT_IMAGE_TOKEN):
            raise ValueError("Number of images does not match number of <image> tokens in prompt")
        images = process_images([load_image_from_base64(i) for i in images], image_processor, model.config)
        images = [i.to(model.device, dtype=torch.float16) for i in images] if type(images) is list else images.to(model.device, dtype=torch.float16)
        replace_token = DEFAULT_IMAGE_TOKEN
        if getattr(model.config, "mm_use_im_start_end", False):
            replace_token = DEFAULT_IM_START_TOKEN + replace_token + DEFAULT_IM_END_TOKEN
        prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, replace_token)
        num_image_tokens = prompt.count(replace_token) * model.get_vision_tower().num_patches
        image_args = {"images": images}
    temperature = float(params.get("temperature", 1.0)); top_p = float(params.get("top_p", 1.0))
    max_context_length = getattr(model.config, "max_position_embeddings", 2048)
    max_new_tokens = min(int(params.get("max_new_tokens", 256)), 1024)
    stop_str = params.get("stop")
    do_sample = temperature > 0.001
    input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).to(model.device)
    stopping_criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
    streamer = TextIteratorStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True, timeout=streamer_timeout)
    max_new_tokens = min(max_new_tokens, max_context_length - input_ids.shape[-1] - num_image_tokens)
    if max_new_tokens < 1:
        yield {"text": ori_prompt + "Exceeds max token length. Please start a new conversation, thanks.", "error_code": 0}
        return
    thread = Thread(target=model.generate, kwargs=dict(inputs=input_ids, do_sample=do_sample, temperature=temperature, top_p=top_p,
                                                       max_new_tokens=max_new_tokens, streamer=streamer,
                                                       stopping_criteria=[stopping_criteria], use_cache=True, **image_args))
    thread.start()
    generated_text = ori_prompt
    for new_text in streamer:
        generated_text += new_text
        if generated_text.endswith(stop_str):
            generated_text = generated_text[:-len(stop_str)]
        yield {"text": generated_text, "error_code": 0}
    thread.join()


def make_worker_app(tokenizer, model, image_processor, limit_model_concurrency=5):
    """model_worker.py:194-244: generate_stream_gate (every exception -> {"error_code": 1}) behind the asyncio semaphore."""
    from fastapi import BackgroundTasks, FastAPI, Request
    from fastapi.responses import StreamingResponse
    app = FastAPI()
    state = {"sem": None, "counter": 0}
    server_error_msg = "**NETWORK ERROR DUE TO HIGH TRAFFIC. PLEASE REGENERATE OR REFRESH THIS PAGE.**"      # llava/utils.py:12

    def gate(params):
        try:
            for x in generate_stream(tokenizer, model, image_processor, params):
                yield json.dumps(x).encode() + b"\0"
        except Exception as e:  # noqa: BLE001  (ValueError / CudaError / anything: model_worker.py:194-218)
            print("Caught error:", repr(e))
            yield json.dumps({"text": server_error_msg, "error_code": 1}).encode() + b"\0"

    @app.post("/worker_generate_stream")
    async def worker_generate_stream(request: Request):
        state["counter"] += 1
        params = await request.json()
        if state["sem"] is None:
            state["sem"] = asyncio.Semaphore(limit_model_concurrency)
        await state["sem"].acquire()
        bg = BackgroundTasks()
        bg.add_task(state["sem"].release)
        return StreamingResponse(gate(params), background=bg)

    @app.post("/worker_get_status")
    async def worker_get_status(request: Request):
        return {"model_names": ["llava-scripted"], "speed": 1, "queue_length": 0}

    return app


def serve_in_thread(app, port, host="127.0.0.1"):
    import uvicorn
    server = uvicorn.Server(uvicorn.Config(app, host=host, port=port, log_level="warning"))
    t = threading.Thread(target=server.run, daemon=True)
    t.start()
    for _ in range(200):
        if server.started:
            break
        time.sleep(0.05)
    return server, t
