"""llava_mi355x — MI355X-native LLaVA-Plus multimodal forward path (CLIP ViT-L/14 -> mm_projector -> LLaMA/Vicuna
decoder with KV cache) behind the reference's Python model API.  Compute = hand-written HIP kernels for gfx950 in
libllava_mi355x.so (C ABI: include/llava_mi355x.h); this package is the host-side mirror of the reference interface.

Importing the package loads the shared library and fails loudly if it is missing — there is no CPU fallback."""
from . import _C  # noqa: F401  (raises ImportError with build instructions when the extension is absent)
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX  # noqa: F401


def __getattr__(name):
    if name in ("LlavaConfig", "LlavaLlamaForCausalLM", "LlavaLlamaModel", "LmxKVCache"):
        from . import model
        return getattr(model, name)
    if name == "load_pretrained_model":
        from .builder import load_pretrained_model
        return load_pretrained_model
    raise AttributeError(name)
