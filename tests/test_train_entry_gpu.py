"""The training entry (llava_mi355x/train_entry.py: train) end to end on the GPU, tiny geometry: json records + images on disk -> LazySupervisedDataset ->
DataCollatorForSupervisedDataset -> frozen tower -> TrainStep (fp32 engine) -> checkpoint, for the two stages of the reference's recipe
(scripts/pretrain.sh: --version plain --tune_mm_mlp_adapter True ; scripts/finetune.sh: --version v1, everything but the tower trains)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _world(tmp_path, n=8):
    from PIL import Image
    from transformers import CLIPImageProcessor
    from synthetic import build as harness, recipes as synth
    from tok_util import build_tokenizer
    cfg = synth.CONFIGS["tiny"]
    wnp = synth.make_weights(cfg, 0)
    lc, _ = harness.hf_configs(cfg)
    model = harness.build_model(cfg, dtype=torch.float32, weights=wnp)
    tower = model.get_vision_tower()
    tower.image_processor = CLIPImageProcessor(size={"shortest_edge": cfg.v_image_size}, crop_size={"height": cfg.v_image_size, "width": cfg.v_image_size})
    rng = np.random.RandomState(0)
    (tmp_path / "img").mkdir()
    records = []
    answers = ["The image shows a dog on the grass.", "The object is in the picture, as the tool outputs show.", "jumps over the lazy dog", "a red ball"]
    for i in range(n):
        Image.fromarray(rng.randint(0, 255, (48 + 4 * i, 64, 3), dtype=np.uint8), "RGB").save(tmp_path / "img" / f"{i}.png")
        records.append({"id": str(i), "image": f"{i}.png",
                        "conversations": [{"from": "human", "value": "<image>\nWhat is shown in the image?"}, {"from": "gpt", "value": answers[i % len(answers)]}]})
    (tmp_path / "data.json").write_text(json.dumps(records))
    weights = {(("model." + k) if k.startswith("mm_projector.") else k): torch.from_numpy(np.array(v)) for k, v in wnp.items()}       # checkpoint (HF) names
    tok = build_tokenizer(model_max_length=128, vocab_size=400)
    data_args = types.SimpleNamespace(data_path=str(tmp_path / "data.json"), image_folder=str(tmp_path / "img"), image_aspect_ratio="pad", lazy_preprocess=True,
                                      is_multimodal=False)
    return cfg, lc, weights, tok, tower, data_args, model


def _args(D, tmp_path, **kw):
    t = D.TrainingArguments(output_dir=str(tmp_path / "out"), bf16=False, per_device_train_batch_size=2, num_train_epochs=4, learning_rate=2e-3, warmup_ratio=0.0,
                            lr_scheduler_type="constant", model_max_length=128, logging_steps=1)
    for k, v in kw.items():
        setattr(t, k, v)
    return t


def test_finetune_stage_trains_everything_but_the_tower(cuda, tmp_path):
    from llava_mi355x import conversation as C, train_data as D, train_entry as E
    cfg, lc, weights, tok, tower, data_args, model = _world(tmp_path)
    before = {k: v.clone() for k, v in weights.items()}
    margs = D.ModelArguments(model_name_or_path="synthetic", version="v1", vision_tower="synthetic-clip", mm_vision_select_layer=cfg.mm_vision_select_layer,
                             mm_projector_type=cfg.mm_projector_type, mm_use_im_start_end=False, mm_use_im_patch_token=False)
    keep = C.default_conversation
    try:
        out = E.train(margs, data_args, _args(D, tmp_path), config=lc, weights=weights, tokenizer=tok, vision_tower=tower)
    finally:
        C.default_conversation = keep
    assert out["steps"] == 16 and len(out["losses"]) == 16 and out["frozen"] == []
    assert all(np.isfinite(out["losses"])) and np.mean(out["losses"][-4:]) < 0.6 * np.mean(out["losses"][:4]), out["losses"]
    saved = torch.load(out["checkpoint"])
    assert os.path.basename(out["checkpoint"]) == "pytorch_model.bin"
    llm = {k for k in before if "vision" not in k}
    assert set(saved) == llm
    changed = [k for k in llm if not torch.equal(saved[k].float(), before[k].float())]
    assert set(changed) == llm                                   # every tensor of the language model and the projector moved
    assert os.path.exists(tmp_path / "out" / "config.json") and json.load(open(tmp_path / "out" / "trainer_state.json"))["global_step"] == 16
    assert data_args.is_multimodal and lc.tokenizer_model_max_length == 128 and lc.image_aspect_ratio == "pad"


def test_pretrain_stage_trains_the_projector_only_and_its_file_loads_back(cuda, tmp_path):
    from llava_mi355x import conversation as C, train_data as D, train_entry as E
    cfg, lc, weights, tok, tower, data_args, model = _world(tmp_path)
    before = {k: v.clone() for k, v in weights.items()}
    margs = D.ModelArguments(model_name_or_path="synthetic", version="plain", vision_tower="synthetic-clip", mm_vision_select_layer=cfg.mm_vision_select_layer,
                             mm_projector_type=cfg.mm_projector_type, tune_mm_mlp_adapter=True, mm_use_im_start_end=False, mm_use_im_patch_token=False)
    keep = C.default_conversation
    try:
        out = E.train(margs, data_args, _args(D, tmp_path, num_train_epochs=3, group_by_modality_length=True), config=lc, weights=weights, tokenizer=tok,
                      vision_tower=tower)                              # (+ the length-grouped sample order of scripts/finetune*.sh)
    finally:
        C.default_conversation = keep
    assert out["steps"] == 12 and out["losses"][-1] < out["losses"][0]
    state = out["state"]
    for k, v in state.items():
        name = ("model." + k) if k.startswith("mm_projector.") else k
        moved = not torch.equal(v.float().cpu(), before[name].float())
        assert moved == ("mm_projector" in k), k                # the language model's bits are untouched, the projector's moved
    assert os.path.basename(out["checkpoint"]) == "mm_projector.bin"
    saved = torch.load(out["checkpoint"])
    assert set(saved) == {k for k in before if "mm_projector" in k}
    # stage 2 starts from it: pretrain_mm_mlp_adapter
    fresh = {k: v.clone() for k, v in before.items() if "mm_projector" not in k and "vision" not in k}
    margs2 = D.ModelArguments(version="v1", vision_tower="synthetic-clip", mm_projector_type=cfg.mm_projector_type, pretrain_mm_mlp_adapter=out["checkpoint"])
    E.initialize_vision_modules(lc, fresh, margs2, tower.hidden_size)
    for k, v in saved.items():
        assert torch.equal(fresh[k].float(), v.float())
