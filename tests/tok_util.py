"""A small LLaMA-shaped tokenizer for the training-data tests (no tokenizer files exist in this image): byte-pair merges trained on a fixed in-file corpus,
LLaMA's normaliser (a "▁" in front, spaces -> "▁", no pre-tokeniser), BOS in front of every encoding, </s> / <s> / <unk> as special tokens, pad = unk as the
reference's training entry sets it (llava/train/train.py:893-900), right padding.  Deterministic: same vocabulary on every run."""
CORPUS = [
    "A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, detailed, and polite answers to the user's questions.",
    "A chat between a curious human and an artificial intelligence assistant. The assistant gives helpful, detailed, and polite answers to the human's questions.",
    "USER: What is shown in the image? ASSISTANT: The image shows a dog on the grass near a red ball.",
    "USER: Where is the object? ASSISTANT: The object is in the picture, as the tool outputs show.",
    "USER: Please segment the object. ASSISTANT: calling the tool", "grounding_dino model outputs: boxes logits phrases",
    "Please summarize the model outputs and answer my first question", "thoughts actions value API_name API_params caption",
    "You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, and assist the user with a variety of tasks using natural language.",
    "[INST] <<SYS>> <</SYS>> [/INST] ### Human: ### Assistant: <|im_start|>user <|im_start|>assistant <|im_end|> system",
    "the quick brown fox jumps over the lazy dog 0 1 2 3 4 5 6 7 8 9 . , : ; ! ? ( ) [ ] { } \" ' / \\ _ - + = < > \n",
]


def build_tokenizer(model_max_length: int = 2048, vocab_size: int = 400):
    from tokenizers import Tokenizer, decoders, models, normalizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE(unk_token="<unk>", fuse_unk=True, byte_fallback=False))
    tok.normalizer = normalizers.Sequence([normalizers.Prepend("▁"), normalizers.Replace(" ", "▁")])
    tok.decoder = decoders.Sequence([decoders.Replace("▁", " "), decoders.Fuse(), decoders.Strip(" ", 1, 0)])
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=["<unk>", "<s>", "</s>"], show_progress=False,
                                  initial_alphabet=sorted(set("".join(CORPUS) + "▁🤔🚀👉")))
    tok.train_from_iterator(CORPUS, trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", pair="<s> $A <s> $B", special_tokens=[("<s>", tok.token_to_id("<s>"))])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", model_max_length=model_max_length,
                                   padding_side="right")
    fast.pad_token = fast.unk_token
    return fast
