"""Training data pipeline of visual instruction tuning: conversations + image -> (input_ids, labels, image) samples -> padded batches
(BASELINE config 5: the data format on the input side of the training step, llava_mi355x/train.py).

Mirror of the data half of llava/train/train.py — same names, same arguments, same tensors out:

    ModelArguments / DataArguments / TrainingArguments     :59-111   (the fields the entry glue and the data side read)
    preprocess_multimodal                                   :316-338  image placeholder to the front of its turn, optional <im_start>/<im_end> wrap
    preprocess_v1 / preprocess_llama_2                      :341-497  template prompt, tokenise, mask everything but the assistant's answers
    preprocess_mpt / preprocess_plain / preprocess          :500-642  (+ the "### role:" fallback with _tokenize_fn / _mask_targets / _add_speaker_and_signal :254-313)
    LazySupervisedDataset                                   :645-746  json records; image load (several folders), pad-to-square, CLIP preprocessing
    DataCollatorForSupervisedDataset                        :749-777  right padding, truncation to model_max_length, attention mask, image stack
    make_supervised_data_module                             :783-800  several json files, comma separated
    split_to_even_chunks / get_(modality_)length_grouped_indices / LengthGroupedSampler     llava/train/llava_trainer.py:38-139  (--group_by_modality_length)

`preprocess` first folds the LLaVA-Plus tool-use fields (thoughts / actions / value) into the answer text (llava/mm_utils.py:117-154).  Label masking is integer
work and must be bit-exact: tests/test_train_data_vs_reference.py runs these functions and the reference's own (imported from the reference tree in the build
container) on the same conversations with the same tokenizer and compares ids and labels with array_equal, for every template the reference's scripts use.
Host code: nothing here touches the GPU; the tensors it makes feed TrainStep.step (input_ids, labels, attention_mask) and the frozen tower (images).
"""
from __future__ import annotations

import copy
import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import conversation as conversation_lib
from .constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IGNORE_INDEX
from .mm_utils import expand2square, reorganize_source_for_tool_use_batch, tokenizer_image_token


@dataclass
class ModelArguments:
    model_name_or_path: Optional[str] = field(default="facebook/opt-125m")
    version: Optional[str] = field(default="v0")
    freeze_backbone: bool = field(default=False)
    tune_mm_mlp_adapter: bool = field(default=False)
    vision_tower: Optional[str] = field(default=None)
    mm_vision_select_layer: Optional[int] = field(default=-1)
    pretrain_mm_mlp_adapter: Optional[str] = field(default=None)
    mm_projector_type: Optional[str] = field(default="linear")
    mm_use_im_start_end: bool = field(default=False)
    mm_use_im_patch_token: bool = field(default=True)
    mm_vision_select_feature: Optional[str] = field(default="patch")


@dataclass
class DataArguments:
    data_path: str = field(default=None, metadata={"help": "Path to the training data (several json files: comma separated)."})
    lazy_preprocess: bool = False
    is_multimodal: bool = False
    image_folder: Optional[str] = field(default=None)
    image_aspect_ratio: str = "square"


@dataclass
class TrainingArguments:
    """The fields of the reference's TrainingArguments (a transformers.TrainingArguments subclass, train.py:83-111) that this package's training entry reads;
    the HF Trainer / DeepSpeed / bitsandbytes / LoRA knobs have no counterpart here (out of scope)."""
    output_dir: str = "./checkpoints"
    model_max_length: int = field(default=512, metadata={"help": "Maximum sequence length. Sequences will be right padded (and possibly truncated)."})
    per_device_train_batch_size: int = 16
    gradient_accumulation_steps: int = 1
    learning_rate: float = 2e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    warmup_ratio: float = 0.03
    lr_scheduler_type: str = "cosine"
    num_train_epochs: float = 1.0
    max_steps: int = -1
    bf16: bool = True
    fp16: bool = False
    gradient_checkpointing: bool = False
    freeze_mm_mlp_adapter: bool = False
    mm_projector_lr: Optional[float] = None
    group_by_modality_length: bool = False
    seed: int = 42
    save_steps: int = 0
    logging_steps: int = 1


# ---- prompt construction + label masking -------------------------------------------------------------------------------------------------------------
def preprocess_multimodal(sources: Sequence[Sequence[dict]], data_args) -> Sequence[Sequence[dict]]:
    """Every turn that mentions the image gets the placeholder in front ("<image>\\n" + the rest); with mm_use_im_start_end the placeholder is wrapped."""
    if not data_args.is_multimodal:
        return sources
    mmtag = "mmtag" in conversation_lib.default_conversation.version
    wrapped = DEFAULT_IMAGE_TOKEN
    if getattr(data_args, "mm_use_im_start_end", False):
        wrapped = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN
    for source in sources:
        for turn in source:
            text = turn["value"]
            if DEFAULT_IMAGE_TOKEN in text:
                text = (DEFAULT_IMAGE_TOKEN + "\n" + text.replace(DEFAULT_IMAGE_TOKEN, "").strip()).strip()
                if mmtag:
                    text = text.replace(DEFAULT_IMAGE_TOKEN, "<Image>" + DEFAULT_IMAGE_TOKEN + "</Image>")
            turn["value"] = text.replace(DEFAULT_IMAGE_TOKEN, wrapped)
    return sources


def _template_prompts(sources) -> List[str]:
    """Each record through the current template: roles checked to alternate from the human side, a leading non-human turn dropped."""
    conv = conversation_lib.default_conversation.copy()
    role_of = {"human": conv.roles[0], "gpt": conv.roles[1]}
    prompts = []
    for i, source in enumerate(sources):
        if role_of[source[0]["from"]] != conv.roles[0]:
            source = source[1:]
        conv.messages = []
        for j, turn in enumerate(source):
            role = role_of[turn["from"]]
            assert role == conv.roles[j % 2], f"{i}"
            conv.append_message(role, turn["value"])
        prompts.append(conv.get_prompt())
    return prompts


def _tokenize_prompts(prompts: List[str], tokenizer, has_image: bool) -> torch.Tensor:
    if has_image:
        return torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    return tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True).input_ids


def _n_tokens(text: str, tokenizer, has_image: bool) -> int:
    return len(tokenizer_image_token(text, tokenizer)) if has_image else len(tokenizer(text).input_ids)


def _mask_by_rounds(prompts, targets, tokenizer, has_image, round_sep: str, answer_sep: str):
    """Labels of a two-separator prompt: position 0 (BOS) and, round by round, the instruction part (everything up to and including `answer_sep`) become
    IGNORE_INDEX; what follows the last complete round too.  A prompt whose re-tokenised rounds do not add up to its length is dropped whole (warning), as in
    the reference.  The "- 2" is the reference's: one for the BOS the separate tokenisation adds, one for the separator's trailing piece."""
    for prompt, target in zip(prompts, targets):
        total = int(target.ne(tokenizer.pad_token_id).sum())
        cur = 1
        target[:cur] = IGNORE_INDEX
        for rnd in prompt.split(round_sep):
            if rnd == "":
                break
            parts = rnd.split(answer_sep)
            if len(parts) != 2:
                break
            n_round = _n_tokens(rnd, tokenizer, has_image)
            n_instr = _n_tokens(parts[0] + answer_sep, tokenizer, has_image) - 2
            target[cur: cur + n_instr] = IGNORE_INDEX
            cur += n_round
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total}. (ignored)")


def preprocess_v1(sources, tokenizer, has_image: bool = False) -> Dict:
    conv = conversation_lib.default_conversation
    prompts = _template_prompts(sources)
    input_ids = _tokenize_prompts(prompts, tokenizer, has_image)
    targets = input_ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    _mask_by_rounds(prompts, targets, tokenizer, has_image, round_sep=conv.sep2, answer_sep=conv.sep + conv.roles[1] + ": ")
    return dict(input_ids=input_ids, labels=targets)


def preprocess_llama_2(sources, tokenizer, has_image: bool = False) -> Dict:
    conv = conversation_lib.default_conversation
    prompts = _template_prompts(sources)
    input_ids = _tokenize_prompts(prompts, tokenizer, has_image)
    targets = input_ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2
    _mask_by_rounds(prompts, targets, tokenizer, has_image, round_sep=conv.sep2, answer_sep="[/INST] ")
    return dict(input_ids=input_ids, labels=targets)


def preprocess_mpt(sources, tokenizer) -> Dict:
    conv = conversation_lib.default_conversation
    prompts = _template_prompts(sources)
    input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    targets = input_ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.MPT
    answer_sep = conv.sep + conv.roles[1]
    n_sep = len(tokenizer_image_token(conv.sep, tokenizer))
    for prompt, target in zip(prompts, targets):
        total = int(target.ne(tokenizer.pad_token_id).sum())
        pieces = prompt.split(conv.sep)
        rounds = [conv.sep.join(pieces[:3])]                                    # system + user + assistant
        rounds += [conv.sep.join(pieces[k:k + 2]) for k in range(3, len(pieces), 2)]
        cur = 0
        for rnd in rounds:
            if rnd == "":
                break
            parts = rnd.split(answer_sep)
            if len(parts) != 2:
                break
            n_instr = len(tokenizer_image_token(parts[0] + answer_sep, tokenizer))
            target[cur: cur + n_instr] = IGNORE_INDEX
            cur += len(tokenizer_image_token(rnd, tokenizer)) + n_sep
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total}. (ignored)")
    return dict(input_ids=input_ids, labels=targets)


def preprocess_plain(sources, tokenizer) -> Dict:
    """Feature-alignment pre-training (`--version plain`): "<image>" + caption + "\\n", everything but the caption masked."""
    prompts = []
    for source in sources:
        assert len(source) == 2
        assert DEFAULT_IMAGE_TOKEN in source[0]["value"]
        source[0]["value"] = DEFAULT_IMAGE_TOKEN
        prompts.append(source[0]["value"] + source[1]["value"] + conversation_lib.default_conversation.sep)
    input_ids = [tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts]
    targets = copy.deepcopy(input_ids)
    for target, source in zip(targets, sources):
        target[: len(tokenizer_image_token(source[0]["value"], tokenizer))] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


def _tokenize_fn(strings: Sequence[str], tokenizer) -> Dict:
    toks = [tokenizer(s, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True) for s in strings]
    ids = [t.input_ids[0] for t in toks]
    lens = [t.input_ids.ne(tokenizer.pad_token_id).sum().item() for t in toks]
    return dict(input_ids=ids, labels=ids, input_ids_lens=lens, labels_lens=lens)


def _mask_targets(target, tokenized_lens, speakers) -> None:
    cur = tokenized_lens[0]
    target[:cur] = IGNORE_INDEX
    for n, speaker in zip(tokenized_lens[1:], speakers):
        if speaker == "human":
            target[cur + 2: cur + n] = IGNORE_INDEX
        cur += n


def _add_speaker_and_signal(header, source, get_conversation=True):
    begin, end = "### ", "\n"
    roles = conversation_lib.default_conversation.roles
    text = header
    for turn in source:
        who = turn["from"].lower()
        name = roles[0] if who == "human" else roles[1] if who == "gpt" else "unknown"
        turn["value"] = begin + name + ": " + turn["value"] + end
        if get_conversation:
            text += turn["value"]
    return text + begin


def preprocess(sources, tokenizer, has_image: bool = False) -> Dict:
    """sources: a list of conversations (lists of {"from", "value", optional "thoughts" / "actions"}).  Dispatch on the current template."""
    sources = reorganize_source_for_tool_use_batch(sources)
    conv = conversation_lib.default_conversation
    if conv.sep_style == conversation_lib.SeparatorStyle.PLAIN:
        return preprocess_plain(sources, tokenizer)
    if conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2:
        return preprocess_llama_2(sources, tokenizer, has_image=has_image)
    if conv.version.startswith("v1"):
        return preprocess_v1(sources, tokenizer, has_image=has_image)
    if conv.version == "mpt":
        return preprocess_mpt(sources, tokenizer)
    # "### role: text" prompts of the v0 templates
    header = f"{conv.system}\n\n"
    prompts = [_add_speaker_and_signal(header, source) for source in sources]
    if has_image:
        input_ids = [tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts]
    else:
        input_ids = _tokenize_fn(prompts, tokenizer)["input_ids"]
    targets = copy.deepcopy(input_ids)
    for target, source in zip(targets, sources):
        texts = [header] + [t["value"] for t in source]
        if has_image:
            lens = [len(tokenizer_image_token(t, tokenizer)) for t in texts]
        else:
            lens = _tokenize_fn(texts, tokenizer)["input_ids_lens"]
        _mask_targets(target, lens, [t["from"] for t in source])
    return dict(input_ids=input_ids, labels=targets)


# ---- dataset + collator ----------------------------------------------------------------------------------------------------------------------------------
class LazySupervisedDataset(torch.utils.data.Dataset):
    """json records {"id", "image"?, "conversations": [...]}: tokenised and masked when a sample is asked for."""

    def __init__(self, data_path: str, tokenizer, data_args):
        super().__init__()
        with open(data_path, "r") as f:
            self.list_data_dict = json.load(f)
        self.tokenizer = tokenizer
        self.data_args = data_args

    def __len__(self):
        return len(self.list_data_dict)

    @property
    def lengths(self) -> List[int]:
        return [sum(len(t["value"].split()) for t in s["conversations"]) + (128 if "image" in s else 0) for s in self.list_data_dict]

    @property
    def modality_lengths(self) -> List[int]:
        out = []
        for s in self.list_data_dict:
            n = sum(len(t["value"].split()) for t in s["conversations"])
            out.append(n if "images" in s else -n)                             # (the reference tests the key "images" here: every sample counts as text-only)
        return out

    def load_image(self, image_file: str, image_folder: str):
        """`image_folder` may name several folders, comma separated: the first one that holds the file wins."""
        from PIL import Image
        if "," not in image_folder:
            return Image.open(os.path.join(image_folder, image_file)).convert("RGB")
        for d in image_folder.split(","):
            path = os.path.join(d.strip(), image_file)
            if os.path.exists(path):
                return Image.open(path).convert("RGB")
        raise ValueError("Unknow_file: {}".format(image_file))

    def __getitem__(self, i) -> Dict[str, torch.Tensor]:
        record = self.list_data_dict[i]
        sources = [record] if isinstance(i, int) else record
        assert len(sources) == 1, "one record per index"
        has_image = "image" in sources[0]
        convs = copy.deepcopy([s["conversations"] for s in sources])
        if has_image:
            processor = self.data_args.image_processor
            image = self.load_image(record["image"], self.data_args.image_folder)
            if self.data_args.image_aspect_ratio == "pad":
                image = expand2square(image, tuple(int(x * 255) for x in processor.image_mean))
            image = processor.preprocess(image, return_tensors="pt")["pixel_values"][0]
            convs = preprocess_multimodal(convs, self.data_args)
        data = preprocess(convs, self.tokenizer, has_image=has_image)
        if isinstance(i, int):
            data = dict(input_ids=data["input_ids"][0], labels=data["labels"][0])
        if has_image:
            data["images"] = image
        elif self.data_args.is_multimodal:
            crop = self.data_args.image_processor.crop_size                     # a text-only sample of a multimodal run carries a blank image
            data["images"] = torch.zeros(3, crop["height"], crop["width"])
        return data


@dataclass
class DataCollatorForSupervisedDataset:
    """Right-pad ids with pad_token_id and labels with IGNORE_INDEX, cut to model_max_length, mask = ids != pad; images stacked when their shapes agree."""
    tokenizer: object

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        pad = self.tokenizer.pad_token_id
        cap = self.tokenizer.model_max_length
        ids = torch.nn.utils.rnn.pad_sequence([x["input_ids"] for x in instances], batch_first=True, padding_value=pad)[:, :cap]
        labels = torch.nn.utils.rnn.pad_sequence([x["labels"] for x in instances], batch_first=True, padding_value=IGNORE_INDEX)[:, :cap]
        batch = dict(input_ids=ids, labels=labels, attention_mask=ids.ne(pad))
        if "images" in instances[0]:
            images = [x["images"] for x in instances]
            same = all(im is not None and im.shape == images[0].shape for im in images)
            batch["images"] = torch.stack(images) if same else images
        return batch


# ---- sample order: batches of similar length (llava/train/llava_trainer.py:38-139, `--group_by_modality_length True` in scripts/finetune*.sh) -------------------
def split_to_even_chunks(indices, lengths, num_chunks):
    """`indices` (sorted longest first by the caller) dealt into num_chunks chunks of equal size and roughly equal total length: each index goes to the chunk
    that is shortest so far and not yet full.  When the count does not divide: a plain stride."""
    if len(indices) % num_chunks != 0:
        return [indices[i::num_chunks] for i in range(num_chunks)]
    per_chunk = len(indices) // num_chunks
    chunks = [[] for _ in range(num_chunks)]
    load = [0] * num_chunks
    for idx in indices:
        k = load.index(min(load))
        chunks[k].append(idx)
        load[k] += lengths[idx]
        if len(chunks[k]) == per_chunk:
            load[k] = float("inf")
    return chunks


def get_length_grouped_indices(lengths, batch_size, world_size, generator=None, merge=True):
    """A random permutation cut into megabatches of world_size x batch_size samples; inside each, longest first, dealt to the ranks by split_to_even_chunks.
    (torch's generator: a distributed run seeds it identically on every rank.)"""
    perm = torch.randperm(len(lengths), generator=generator)
    mega = world_size * batch_size
    out = []
    for i in range(0, len(lengths), mega):
        block = sorted(perm[i: i + mega].tolist(), key=lambda j: lengths[j], reverse=True)
        for chunk in split_to_even_chunks(block, lengths, world_size):
            out.extend(chunk)
    return out


def get_modality_length_grouped_indices(lengths, batch_size, world_size, generator=None):
    """lengths > 0: samples with an image, < 0: text-only (LazySupervisedDataset.modality_lengths).  Megabatches hold ONE modality each (grouped by length inside);
    their order is shuffled; the two left-over part-megabatches form a last, mixed one."""
    assert all(l != 0 for l in lengths), "Should not have zero length."
    if all(l > 0 for l in lengths) or all(l < 0 for l in lengths):
        return get_length_grouped_indices(lengths, batch_size, world_size, generator=generator)
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    mega = world_size * batch_size

    def grouped(pairs):
        idx, lens = zip(*pairs)
        order = [idx[j] for j in get_length_grouped_indices(lens, batch_size, world_size, generator=None)]
        return [order[i: i + mega] for i in range(0, len(order), mega)]
    mm_mb, lang_mb = grouped(mm), grouped(lang)
    rest = mm_mb[-1] + lang_mb[-1]
    full = mm_mb[:-1] + lang_mb[:-1]
    full = [full[i] for i in torch.randperm(len(full), generator=generator)]
    if len(rest) > 0:
        full.append(sorted(rest))
    return [i for mb in full for i in mb]


class LengthGroupedSampler(torch.utils.data.Sampler):
    """Index order of an epoch: samples of similar length share a batch (less padding), optionally one modality per megabatch."""

    def __init__(self, batch_size: int, world_size: int, lengths: Optional[List[int]] = None, generator=None, group_by_modality: bool = False):
        if lengths is None:
            raise ValueError("Lengths must be provided.")
        self.batch_size, self.world_size, self.lengths, self.generator, self.group_by_modality = batch_size, world_size, lengths, generator, group_by_modality

    def __len__(self):
        return len(self.lengths)

    def __iter__(self):
        f = get_modality_length_grouped_indices if self.group_by_modality else get_length_grouped_indices
        return iter(f(self.lengths, self.batch_size, self.world_size, generator=self.generator))


def build_dataset(data_args, tokenizer, dataset_cls):
    return dataset_cls(tokenizer=tokenizer, data_path=data_args.data_path, data_args=data_args)


def make_supervised_data_module(tokenizer, data_args) -> Dict:
    """Dataset (the json files of data_args.data_path, comma separated, concatenated) + collator.  (The reference's function builds the same ConcatDataset
    and then falls off its end without returning it, train.py:783-800; the dict here is what its caller unpacks into the trainer.)"""
    paths = [p.strip() for p in data_args.data_path.split(",") if p.strip() != ""]
    parts = []
    for path in paths:
        assert os.path.exists(path), f"{path} does not exist"
        one = copy.copy(data_args)
        one.data_path = path
        parts.append(build_dataset(one, tokenizer, LazySupervisedDataset))
    return dict(train_dataset=torch.utils.data.ConcatDataset(parts), eval_dataset=None, data_collator=DataCollatorForSupervisedDataset(tokenizer=tokenizer))
