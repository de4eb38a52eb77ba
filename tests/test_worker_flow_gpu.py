"""The serving call sequence of the reference's model worker, end to end on this build.

`/root/reference` is not present on the GPU box, so the unmodified worker cannot be imported here; this test performs
the same calls in the same order as `ModelWorker.__init__` + `generate_stream` (llava/serve/model_worker.py:65-66,
122-192): load_pretrained_model -> base64 image -> process_images -> fp16 tensor on model.device -> prompt rewrite ->
tokenizer_image_token -> KeywordsStoppingCriteria + TextIteratorStreamer -> model.generate on a fresh Thread with exactly
the worker's keyword arguments -> text chunks read from the streamer.  Three requests run concurrently (the worker's
--limit-model-concurrency model), once on independent threads and once with continuous batching switched on through the
LLAVA_MI355X_BATCH environment variable; greedy requests must produce the same text either way."""
import base64
import io
import json
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _b64_image(seed, size=(70, 50)):
    from PIL import Image
    rng = np.random.RandomState(seed)
    img = Image.fromarray(rng.randint(0, 255, size=(size[1], size[0], 3), dtype=np.uint8), "RGB")
    buf = io.BytesIO(); img.save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode()


def _generate_stream(tokenizer, model, image_processor, params):
    """Same steps as the worker's generate_stream; yields the JSON chunks it would send."""
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
    streamer = TextIteratorStreamer(tokenizer, skip_prompt=True, skip_special_tokens=True, timeout=60)
    max_new_tokens = min(max_new_tokens, max_context_length - input_ids.shape[-1] - num_image_tokens)
    if max_new_tokens < 1:
        yield json.dumps({"text": ori_prompt + "Exceeds max token length. Please start a new conversation, thanks.", "error_code": 0})
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
        yield json.dumps({"text": generated_text, "error_code": 0})
    thread.join()


def _serve(tokenizer, model, image_processor, requests):
    """every request on its own thread, like the worker's FastAPI handlers."""
    out, errors = [None] * len(requests), []

    def run(i):
        try:
            chunks = [json.loads(c) for c in _generate_stream(tokenizer, model, image_processor, requests[i])]
            out[i] = chunks[-1]["text"] if chunks else ""
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(len(requests))]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errors, errors
    return out


def test_worker_generate_stream_flow(cuda, tmp_path, monkeypatch):
    from ckpt_util import write_clip, write_llava
    from llava_mi355x.builder import load_pretrained_model
    from synthetic import recipes as synth
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    clip_dir = str(tmp_path / "clip-tiny"); write_clip(clip_dir, cfg, wnp, "4.31")
    ckpt = str(tmp_path / "llava-tiny-7b"); write_llava(ckpt, cfg, wnp, clip_dir)
    reqs = [
        {"prompt": "w1 w2 <image>\nw3 w4 w5", "images": [_b64_image(1)], "temperature": 0.0, "max_new_tokens": 12, "stop": "</s>"},
        {"prompt": "w7 <image>\nw9", "images": [_b64_image(2, (40, 64))], "temperature": 0.0, "max_new_tokens": 7, "stop": "</s>"},
        {"prompt": "w11 w12 w13 w14", "temperature": 0.0, "max_new_tokens": 9, "stop": "</s>"},                      # text only
        {"prompt": "w2 <image>\nw3", "images": [_b64_image(3)], "temperature": 0.7, "top_p": 0.9, "max_new_tokens": 6, "stop": "</s>"},
    ]
    monkeypatch.delenv("LLAVA_MI355X_BATCH", raising=False)
    tokenizer, model, image_processor, context_len = load_pretrained_model(ckpt, None, "llava-tiny-7b", torch_dtype=torch.float32)
    model.config.image_aspect_ratio = "pad"                   # the v1.5 serving setting: expand2square with the mean colour
    assert model._batcher is None
    plain = _serve(tokenizer, model, image_processor, reqs)
    for r, text in zip(reqs, plain):
        assert text.startswith(r["prompt"])
    # mismatch between images and <image> markers is the worker's ValueError
    with pytest.raises(ValueError):
        list(_generate_stream(tokenizer, model, image_processor, {"prompt": "no marker", "images": [_b64_image(4)], "stop": "</s>"}))
    # a prompt that leaves no room answers without generating
    long_req = {"prompt": " ".join(["w1"] * 300), "temperature": 0.0, "stop": "</s>"}
    chunks = list(_generate_stream(tokenizer, model, image_processor, long_req))
    assert len(chunks) == 1 and "Exceeds max token length" in chunks[0]

    monkeypatch.setenv("LLAVA_MI355X_BATCH", "4")
    tokenizer2, model2, image_processor2, _ = load_pretrained_model(ckpt, None, "llava-tiny-7b", torch_dtype=torch.float32)
    model2.config.image_aspect_ratio = "pad"
    assert model2._batcher is not None
    batched = _serve(tokenizer2, model2, image_processor2, reqs)
    for i in range(3):                                        # greedy requests: identical text with and without batching
        assert batched[i] == plain[i], (i, batched[i], plain[i])
    assert batched[3].startswith(reqs[3]["prompt"])
    # image preprocessing on the GPU (LLAVA_MI355X_DEVICE_PREPROCESS=1): process_images returns the same pixels, so the same text
    monkeypatch.setenv("LLAVA_MI355X_DEVICE_PREPROCESS", "1")
    on_device = _serve(tokenizer2, model2, image_processor2, reqs[:3])
    assert on_device == plain[:3]
    model2.disable_batching()
