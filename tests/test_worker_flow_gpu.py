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
    """ModelWorker.generate_stream, re-enacted (tools/worker_reenactment.py); yields the JSON chunks it would send."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from worker_reenactment import generate_stream
    for chunk in generate_stream(tokenizer, model, image_processor, params):
        yield json.dumps(chunk)


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
