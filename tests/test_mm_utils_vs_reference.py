"""CPU, build container only: the host helpers of this package (llava_mi355x/mm_utils.py, constants.py) against the REFERENCE's
own llava/mm_utils.py imported through oracle/ref_shim.py — same inputs, same outputs.  Skipped where /root/reference does not
exist (the GPU box); the recorded KATs of tests/golden/tokenizer_image_token.json cover that case."""
import base64
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))

from oracle import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")


class FakeTokenizer:
    """The fake tokenizer of SURVEY Appendix B1: ids = [bos] + [10 + ord(c) % 50 for c in text]."""
    bos_token_id = 1

    def __call__(self, text):
        class R: pass
        r = R(); r.input_ids = [1] + [10 + ord(c) % 50 for c in text]
        return r

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["".join(chr(97 + int(t) % 26) for t in row) for row in ids]

    def decode(self, ids, **kw):
        return "".join(chr(97 + int(t) % 26) for t in ids)


def _img(seed, w, h):
    from PIL import Image
    rng = np.random.RandomState(seed)
    return Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8), "RGB")


@pytest.fixture(scope="module")
def ref():
    return ref_shim.load_reference()


def test_constants_equal(ref):
    from llava_mi355x import constants as C
    for name in ("IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "DEFAULT_IMAGE_TOKEN", "DEFAULT_IMAGE_PATCH_TOKEN", "DEFAULT_IM_START_TOKEN", "DEFAULT_IM_END_TOKEN",
                 "IMAGE_PLACEHOLDER", "WORKER_HEART_BEAT_INTERVAL", "CONTROLLER_HEART_BEAT_EXPIRATION", "LOGDIR"):
        assert getattr(C, name) == getattr(ref.constants, name), name


@pytest.mark.parametrize("prompt", ["AB<image>\nCD", "<image>\nX", "no image", "A<image>B<image>C", "", "<image>", "x<image><image>y"])
def test_tokenizer_image_token_equal(ref, prompt):
    from llava_mi355x import mm_utils as M
    tok = FakeTokenizer()
    assert M.tokenizer_image_token(prompt, tok) == ref.mm_utils.tokenizer_image_token(prompt, tok)
    a = M.tokenizer_image_token(prompt, tok, return_tensors="pt"); b = ref.mm_utils.tokenizer_image_token(prompt, tok, return_tensors="pt")
    assert torch.equal(a, b)
    assert M.tokenizer_image_token(prompt, tok, image_token_index=-7) == ref.mm_utils.tokenizer_image_token(prompt, tok, image_token_index=-7)


@pytest.mark.parametrize("w,h", [(70, 50), (50, 70), (64, 64), (1, 9)])
def test_expand2square_and_base64_equal(ref, w, h):
    from llava_mi355x import mm_utils as M
    im = _img(w * 7 + h, w, h)
    fill = (122, 116, 104)
    assert np.array_equal(np.asarray(M.expand2square(im, fill)), np.asarray(ref.mm_utils.expand2square(im, fill)))
    buf = io.BytesIO(); im.save(buf, format="PNG"); b64 = base64.b64encode(buf.getvalue()).decode()
    assert np.array_equal(np.asarray(M.load_image_from_base64(b64)), np.asarray(ref.mm_utils.load_image_from_base64(b64)))


@pytest.mark.parametrize("aspect", [None, "pad"])
def test_process_images_equal(ref, aspect):
    from transformers import CLIPImageProcessor
    from llava_mi355x import mm_utils as M
    proc = CLIPImageProcessor(size={"shortest_edge": 56}, crop_size={"height": 56, "width": 56})

    class Cfg: image_aspect_ratio = aspect
    ims = [_img(1, 90, 60), _img(2, 40, 77), _img(3, 56, 56)]
    a = M.process_images(ims, proc, Cfg()); b = ref.mm_utils.process_images(ims, proc, Cfg())
    assert type(a) is type(b)
    assert torch.equal(torch.as_tensor(a), torch.as_tensor(b))


def test_model_name_and_stopping_criteria_equal(ref):
    from llava_mi355x import mm_utils as M
    for p in ("/a/b/llava-v1.5-7b", "/x/checkpoint-300", "liuhaotian/llava-v1.5-13b/", "/m/llava-plus/checkpoint-12"):
        assert M.get_model_name_from_path(p) == ref.mm_utils.get_model_name_from_path(p)
    tok = FakeTokenizer()
    prompt_ids = torch.tensor([[1, 5, 9, 12]])
    for kw in (["stop"], ["ab", "xyz"], ["</s>"]):
        mine = M.KeywordsStoppingCriteria(kw, tok, prompt_ids); theirs = ref.mm_utils.KeywordsStoppingCriteria(kw, tok, prompt_ids)
        rng = np.random.RandomState(len(kw))
        for n in range(1, 12):
            out = torch.cat([prompt_ids, torch.from_numpy(rng.randint(0, 26, size=(1, n)))], dim=1)
            assert bool(mine(out, None)) == bool(theirs(out, None)), (kw, n)
