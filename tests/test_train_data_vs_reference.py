"""CPU, build container only: the training-data side of this package (llava_mi355x/conversation.py, train_data.py, mm_utils.reorganize_source_for_tool_use)
against the REFERENCE's own llava/conversation.py and llava/train/train.py, imported from /root/reference through oracle/ref_shim.load_reference_train —
same conversations, same tokenizer (tests/tok_util.py: LLaMA-shaped, deterministic), same tensors out.  Label masks are integer work: array_equal.
Skipped where the reference tree does not exist (the GPU box); tests/test_train_data.py holds the properties that need no reference."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available() or ref_shim.is_sourceless(), reason="reference source tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_shim.load_reference_train()


@pytest.fixture(scope="module")
def tok():
    from tok_util import build_tokenizer
    return build_tokenizer(model_max_length=256)


def conversations():
    """LLaVA-style and LLaVA-Plus-style (thoughts / actions / value) records; single and multi turn; with and without the image placeholder."""
    return [
        [{"from": "human", "value": "<image>\nWhat is shown in the image?"}, {"from": "gpt", "value": "The image shows a dog on the grass."}],
        [{"from": "human", "value": "Where is the object? <image>"},
         {"from": "gpt", "thoughts": "I need a detector.", "actions": [{"API_name": "grounding_dino", "API_params": {"caption": "the object ."}}], "value": "calling the tool"},
         {"from": "human", "value": "grounding_dino model outputs: {'boxes': [[0.12, 0.2, 0.65, 0.7]]}\n\nPlease summarize the model outputs and answer my first question"},
         {"from": "gpt", "thoughts": "The tool has answered.", "actions": [], "value": "The object is in the picture, as the tool outputs show."}],
        [{"from": "gpt", "value": "a leading assistant turn is dropped"}, {"from": "human", "value": "the quick brown fox?"}, {"from": "gpt", "value": "jumps over the lazy dog"}],
        [{"from": "human", "value": "no image here: 1 2 3"}, {"from": "gpt", "value": "4 5 6"}, {"from": "human", "value": "and then?"}, {"from": "gpt", "value": "7 8 9 !"}],
    ]


def set_template(ref, name):
    from llava_mi355x import conversation as C
    ref.conversation.default_conversation = ref.conversation.conv_templates[name]
    C.default_conversation = C.conv_templates[name]


@pytest.fixture(autouse=True)
def restore_template(ref):
    from llava_mi355x import conversation as C
    a, b = ref.conversation.default_conversation, C.default_conversation
    yield
    ref.conversation.default_conversation, C.default_conversation = a, b


def _prompt(c):
    """The prompt, or the exception type where the template cannot render the history (the plain template has no second separator: an answer turn with text
    raises in the reference too)."""
    try:
        return c.get_prompt()
    except Exception as e:  # noqa: BLE001
        return type(e)


def test_templates_and_prompts_equal(ref):
    from llava_mi355x import conversation as C
    assert set(C.conv_templates) == set(ref.conversation.conv_templates)
    msgs = [("q one", "a one"), ("<image>\nq two", "a two"), ("q three", None)]
    for name, rc in ref.conversation.conv_templates.items():
        mc = C.conv_templates[name]
        for f in ("system", "sep", "sep2", "version"):
            assert getattr(mc, f) == getattr(rc, f), (name, f)
        assert tuple(mc.roles) == tuple(rc.roles) and mc.sep_style.name == rc.sep_style.name
        if name in ("default", "v0"):
            continue                                     # the reference's few-shot Vicuna-v0 prompt (two canned exchanges): documented difference
        a, b = mc.copy(), rc.copy()
        for q, ans in msgs:
            for c in (a, b):
                c.append_message(c.roles[0], q); c.append_message(c.roles[1], ans)
        assert _prompt(a) == _prompt(b), name
        # first message as the web client's tuple (text, image, mode): the placeholder moves to the front (or into its own turn for the mmtag templates)
        a, b = mc.copy(), rc.copy()
        for c in (a, b):
            c.append_message(c.roles[0], ("look at this <image> please", object(), "Pad")); c.append_message(c.roles[1], None)
        assert _prompt(a) == _prompt(b), name
    assert C.SeparatorStyle.__members__.keys() == ref.conversation.SeparatorStyle.__members__.keys()


@pytest.mark.parametrize("start_end", [False, True])
@pytest.mark.parametrize("template", ["v1", "v1_mmtag", "plain"])
def test_preprocess_multimodal_equal(ref, template, start_end):
    from llava_mi355x import train_data as D
    set_template(ref, template)

    class Args:
        is_multimodal = True
        mm_use_im_start_end = start_end
    a = D.preprocess_multimodal(copy.deepcopy(conversations()), Args())
    b = ref.train.preprocess_multimodal(copy.deepcopy(conversations()), Args())
    assert a == b
    Args.is_multimodal = False
    src = conversations()
    assert D.preprocess_multimodal(src, Args()) is src


@pytest.mark.parametrize("has_image", [True, False])
@pytest.mark.parametrize("template", ["v1", "llava_v1", "llava_llama_2", "llama_2", "llava_v0", "v0_mmtag", "v1_mmtag"])
def test_preprocess_equal(ref, tok, template, has_image):
    """`preprocess` (tool-use folding + template + tokenisation + label mask) on every template family, one conversation at a time and as a batch."""
    from llava_mi355x import train_data as D
    set_template(ref, template)
    convs = conversations()
    if not has_image:
        convs = [[dict(t, value=t["value"].replace("<image>", "").strip()) if "value" in t else t for t in c] for c in convs]
    groups = [[c] for c in convs]
    if has_image or template in ("llava_v0", "v0_mmtag"):
        groups.append(convs[:1] * 2)                     # (image prompts stack only when equally long; the v0 path returns lists)
    else:
        groups.append(convs)
    for g in groups:
        a = D.preprocess(copy.deepcopy(g), tok, has_image=has_image)
        b = ref.train.preprocess(copy.deepcopy(g), tok, has_image=has_image)
        for key in ("input_ids", "labels"):
            xa, xb = a[key], b[key]
            assert type(xa) is type(xb)
            xa = list(xa) if isinstance(xa, (list, tuple)) else [xa]
            xb = list(xb) if isinstance(xb, (list, tuple)) else [xb]
            assert len(xa) == len(xb)
            for u, v in zip(xa, xb):
                assert u.dtype == v.dtype and torch.equal(u, v), (template, key)
    # the mask really separates prompt from answer for the standard template (not everything dropped as a tokenisation mismatch)
    if template == "v1":
        one = D.preprocess(copy.deepcopy(convs[:1]), tok, has_image=has_image)
        lab = one["labels"][0]
        assert (lab != -100).any() and (lab == -100).any() and int(lab[0]) == -100


@pytest.mark.parametrize("template", ["plain", "mpt"])
def test_preprocess_plain_and_mpt_equal(ref, tok, template):
    from llava_mi355x import train_data as D
    set_template(ref, template)
    convs = conversations()[:1] if template == "plain" else conversations()[:2]
    for g in ([c] for c in convs):
        a = D.preprocess(copy.deepcopy(g), tok, has_image=True)
        b = ref.train.preprocess(copy.deepcopy(g), tok, has_image=True)
        for key in ("input_ids", "labels"):
            for u, v in zip(list(a[key]), list(b[key])):
                assert torch.equal(u, v), (template, key)


def test_dataset_and_collator_equal(ref, tok, tmp_path):
    """json + images on disk -> LazySupervisedDataset samples -> DataCollatorForSupervisedDataset batches: ids, labels, mask and pixel tensors equal the
    reference classes'; several image folders, pad-to-square, a text-only record in a multimodal run, truncation to model_max_length."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    from llava_mi355x import train_data as D
    set_template(ref, "v1")
    rng = np.random.RandomState(0)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    Image.fromarray(rng.randint(0, 255, (40, 64, 3), dtype=np.uint8), "RGB").save(tmp_path / "a" / "one.png")
    Image.fromarray(rng.randint(0, 255, (70, 30, 3), dtype=np.uint8), "RGB").save(tmp_path / "b" / "two.png")
    convs = conversations()
    records = [{"id": "0", "image": "one.png", "conversations": convs[0]}, {"id": "1", "image": "two.png", "conversations": convs[1]},
               {"id": "2", "conversations": [dict(t, value=t["value"]) for t in convs[3]]}]
    path = tmp_path / "data.json"
    path.write_text(json.dumps(records))
    proc = CLIPImageProcessor(size={"shortest_edge": 28}, crop_size={"height": 28, "width": 28})

    def args(aspect):
        class A:
            data_path = str(path)
            image_folder = f"{tmp_path / 'a'}, {tmp_path / 'b'}"
            image_aspect_ratio = aspect
            image_processor = proc
            is_multimodal = True
            mm_use_im_start_end = False
            lazy_preprocess = True
        return A()
    for aspect in ("square", "pad"):
        mine = D.LazySupervisedDataset(str(path), tok, args(aspect))
        theirs = ref.train.LazySupervisedDataset(str(path), tok, args(aspect))
        assert len(mine) == len(theirs) == 3 and mine.lengths == theirs.lengths and mine.modality_lengths == theirs.modality_lengths
        sa, sb = [mine[i] for i in range(3)], [theirs[i] for i in range(3)]
        for x, y in zip(sa, sb):
            assert x.keys() == y.keys()
            for k in x:
                assert torch.equal(x[k], y[k]), (aspect, k)
        small = copy.copy(tok); small.model_max_length = 48        # shorter than the tool conversation: the collator truncates
        for t in (tok, small):
            ba = D.DataCollatorForSupervisedDataset(tokenizer=t)(sa)
            bb = ref.train.DataCollatorForSupervisedDataset(tokenizer=t)(sb)
            assert ba.keys() == bb.keys()
            for k in ba:
                assert torch.equal(ba[k], bb[k]), (aspect, k)
            assert ba["input_ids"].shape[1] <= t.model_max_length and ba["images"].shape == (3, 3, 28, 28)
    # several json files, comma separated
    path2 = tmp_path / "more.json"
    path2.write_text(json.dumps(records[:1]))
    a = args("square"); a.data_path = f"{path}, {path2}"
    mod = D.make_supervised_data_module(tok, a)
    assert len(mod["train_dataset"]) == 4 and isinstance(mod["data_collator"], D.DataCollatorForSupervisedDataset)
    assert torch.equal(mod["train_dataset"][3]["input_ids"], sa[0]["input_ids"])


def test_argument_dataclasses_cover_the_reference_fields(ref):
    import dataclasses
    from llava_mi355x import train_data as D
    for name in ("ModelArguments", "DataArguments"):
        mine = {f.name: f.default for f in dataclasses.fields(getattr(D, name))}
        theirs = {f.name: f.default for f in dataclasses.fields(getattr(ref.train, name))}
        assert mine == theirs, name
    theirs = {f.name for f in dataclasses.fields(ref.train.TrainingArguments)}
    mine = {f.name for f in dataclasses.fields(D.TrainingArguments)}
    # every field here exists (same name) in the reference's HF TrainingArguments subclass; `warmup_ratio` (scripts/*.sh pass --warmup_ratio 0.03) is a
    # transformers 4.31 field that the transformers 5 installed in this image no longer has
    assert mine - {"warmup_ratio"} <= theirs, mine - theirs


def test_length_grouped_sampler_equals_the_reference_functions():
    """llava/train/llava_trainer.py's sampler helpers (the file itself does not import under the installed transformers: its pure functions are compiled from
    its source here) against train_data's, same torch generator: identical index orders for one modality, mixed modalities, and counts that do not divide."""
    import ast
    from llava_mi355x import train_data as D
    src = open(os.path.join(ref_shim.REF_ROOT, "llava", "train", "llava_trainer.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("split_to_even_chunks", "get_modality_length_grouped_indices", "get_length_grouped_indices")]
    assert len(keep) == 3
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "llava_trainer.py", "exec"), ns)
    rng = np.random.RandomState(3)
    for n, bs, ws, mixed in ((64, 4, 2, False), (61, 4, 2, True), (128, 16, 8, True), (37, 2, 4, True), (48, 4, 4, False), (200, 8, 2, True)):
        lens = rng.randint(1, 300, size=n)
        if mixed:
            lens = lens * rng.choice([-1, 1], size=n)
        lens = [int(x) for x in lens]
        for seed in (0, 5):
            ga, gb = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed)
            pos = [abs(x) for x in lens]
            assert D.get_length_grouped_indices(pos, bs, ws, generator=ga) == ns["get_length_grouped_indices"](pos, bs, ws, generator=gb)
            torch.manual_seed(seed); a = D.get_modality_length_grouped_indices(lens, bs, ws, generator=ga)
            torch.manual_seed(seed); b = ns["get_modality_length_grouped_indices"](lens, bs, ws, generator=gb)
            assert a == b and sorted(a) == list(range(n))
            torch.manual_seed(seed); c = list(D.LengthGroupedSampler(bs, ws, lengths=lens, generator=torch.Generator().manual_seed(seed), group_by_modality=True))
            torch.manual_seed(seed); d = ns["get_modality_length_grouped_indices"](lens, bs, ws, generator=torch.Generator().manual_seed(seed))
            assert c == d
