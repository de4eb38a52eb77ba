"""Host-side pre/post-processing that produces the hot path's inputs (boundary-adjacent).

Behavioural mirror of the reference's llava/mm_utils.py (same names, arguments and results) so the serving code that
imports these helpers keeps working when this package stands in for `llava`:
  load_image_from_base64 :12-13 · expand2square :16-27 · process_images :30-44 · tokenizer_image_token :47-67 ·
  get_model_name_from_path :70-76 · KeywordsStoppingCriteria :79-114
Pure host code (PIL / tokenizer); nothing here touches the GPU.
"""
from __future__ import annotations

import base64
from io import BytesIO

import torch

from .constants import IMAGE_TOKEN_INDEX


def load_image_from_base64(image):
    from PIL import Image
    return Image.open(BytesIO(base64.b64decode(image)))


def expand2square(pil_img, background_color):
    """Pad the short side (centred) with `background_color` so the image becomes square; square images pass through."""
    from PIL import Image
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


_DEVICE_MODEL = None     # engine used by process_images when LLAVA_MI355X_DEVICE_PREPROCESS=1 (set by load_pretrained_model)


def set_device_preprocess_model(model) -> None:
    global _DEVICE_MODEL
    _DEVICE_MODEL = model


def process_images_device(images, image_processor, model_cfg, model):
    """process_images on the GPU (csrc/preprocess.hip): each decoded RGB image goes to HBM as uint8 and is resized (Pillow's
    bicubic resampler, bit-exact), center-cropped, rescaled and normalised there.  Returns a float32 CUDA tensor [N, 3, S, S]
    (the worker's `.to(device, dtype=float16)` then is a device-side cast)."""
    import ctypes
    import numpy as np
    from ._C import check, lib, ptr, torch_dtype_code
    pad = getattr(model_cfg, "image_aspect_ratio", None) == "pad"
    S = model.vision_config.image_size
    crop = getattr(image_processor, "crop_size", None) or {}
    size = getattr(image_processor, "size", None) or {}
    if (crop.get("height", S), crop.get("width", S)) != (S, S) or size.get("shortest_edge", S) != S:
        raise ValueError("device preprocessing implements the CLIP recipe with resize == crop == the tower's image size")
    mean = (ctypes.c_float * 3)(*[float(v) for v in image_processor.image_mean])
    std = (ctypes.c_float * 3)(*[float(v) for v in image_processor.image_std])
    out = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=model.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(model.device).cuda_stream)
    for i, im in enumerate(images):
        rgb = torch.from_numpy(np.array(im.convert("RGB"), dtype=np.uint8, order="C")).to(model.device)
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        with torch.cuda.device(model.device):
            check(lib.lmx_preprocess_image(model._h, ptr(rgb), H, W, torch_dtype_code(torch.float32), int(pad), mean, std, ptr(out[i]), stream),
                  "lmx_preprocess_image")
    return out


def process_images(images, image_processor, model_cfg):
    """image_aspect_ratio == 'pad': pad each image to a square with the processor's mean colour, preprocess one by one and
    stack when shapes agree; otherwise hand the whole list to the processor.
    With LLAVA_MI355X_DEVICE_PREPROCESS=1 (and a model loaded by load_pretrained_model) the same result is produced on the GPU."""
    import os
    if _DEVICE_MODEL is not None and os.environ.get("LLAVA_MI355X_DEVICE_PREPROCESS", "0") == "1":
        return process_images_device(images, image_processor, model_cfg, _DEVICE_MODEL)
    if getattr(model_cfg, "image_aspect_ratio", None) != "pad":
        return image_processor(images, return_tensors="pt")["pixel_values"]
    fill = tuple(int(c * 255) for c in image_processor.image_mean)
    out = [image_processor.preprocess(expand2square(im, fill), return_tensors="pt")["pixel_values"][0] for im in images]
    if all(x.shape == out[0].shape for x in out):
        out = torch.stack(out, dim=0)
    return out


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text around every '<image>' and put `image_token_index` between the pieces; exactly one BOS survives
    (the first piece's), per-piece BOS tokens are dropped."""
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split("<image>")]
    has_bos = bool(pieces) and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [pieces[0][0]] if has_bos else []
    for i, piece in enumerate(pieces):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


class KeywordsStoppingCriteria:
    """Stop when the generated tail equals a keyword's token ids or its decoded text contains a keyword.
    Duck-typed like transformers.StoppingCriteria (callable (output_ids, scores) -> bool); batch = all rows stop."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs) -> bool:
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        self.keyword_ids = [k.to(output_ids.device) for k in self.keyword_ids]
        for k in self.keyword_ids:
            if (output_ids[0, -k.shape[0]:] == k).all():
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores, **kwargs) -> bool:
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))


_TOOL_FIELDS = (("thoughts", "thoughts🤔"), ("actions", "actions🚀"), ("value", "value👉"))


def reorganize_source_for_tool_use(source):
    """LLaVA-Plus training records carry an assistant turn as up to three fields — `thoughts`, `actions` (a list of {API_name, API_params}) and `value`;
    the text the model learns is their concatenation, one `"<tag>" <content>\\n` line each, actions as JSON (llava/mm_utils.py:117-149).  Human turns pass
    through; the assistant dicts are rewritten in place (fields popped, `value` = the merged text), as the reference does."""
    import json
    out = []
    for turn in source:
        if turn["from"].lower() != "human":
            merged = ""
            for key, tag in _TOOL_FIELDS:
                if key in turn:
                    content = turn.pop(key)
                    merged += '"{}" {}'.format(tag, json.dumps(content) if key == "actions" else content) + "\n"
            turn["value"] = merged
        out.append(turn)
    return out


def reorganize_source_for_tool_use_batch(sources):
    return [reorganize_source_for_tool_use(s) for s in sources]
