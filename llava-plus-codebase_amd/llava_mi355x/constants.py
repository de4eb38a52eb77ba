"""Names the serving code imports from `llava.constants`, with the values the multimodal path keys on
(reference: llava/constants.py:1-13; equality is checked in tests/test_mm_utils_vs_reference.py)."""

# ids the splice (prepare_inputs_labels_for_multimodal, llava_arch.py:150-187) looks for / writes
IMAGE_TOKEN_INDEX = -200        # placed by tokenizer_image_token wherever the prompt says "<image>"
IGNORE_INDEX = -100             # label of every image position and of padding

# prompt-side strings
DEFAULT_IMAGE_TOKEN = "<image>"
_TAG = "<im_{}>"
DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN = (_TAG.format(k) for k in ("start", "end", "patch"))
IMAGE_PLACEHOLDER = "<image-placeholder>"

# worker / controller plumbing values that live in the same module upstream (seconds, log directory)
WORKER_HEART_BEAT_INTERVAL, CONTROLLER_HEART_BEAT_EXPIRATION = 15, 30
LOGDIR = "."
