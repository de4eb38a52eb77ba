"""Self-generated HF-format LLaVA checkpoints for the loader / worker-flow tests (no network, no real weights):
sharded safetensors with the reference's key names, config.json with the llava fields, a CLIP tower directory, a
sentencepiece LLaMA tokenizer trained on the spot."""
import json
import os

import torch

def write_tokenizer(d):
    import sentencepiece as spm
    os.environ.setdefault("GLOG_minloglevel", "2")
    corpus = os.path.join(d, "corpus.txt")
    with open(corpus, "w") as f:
        for i in range(400):
            f.write(" ".join(f"w{(i * 7 + j) % 97}" for j in range(12)) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, "tokenizer"), vocab_size=300, model_type="bpe",
                                   bos_id=1, eos_id=2, unk_id=0, pad_id=-1, byte_fallback=True, character_coverage=1.0, minloglevel=2)
    os.remove(corpus)
    json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "legacy": True, "add_bos_token": True, "add_eos_token": False},
              open(os.path.join(d, "tokenizer_config.json"), "w"))


def write_clip(d, cfg, wnp, layout):
    from safetensors.torch import save_file
    from transformers import CLIPImageProcessor, CLIPVisionConfig
    os.makedirs(d, exist_ok=True)
    CLIPVisionConfig(hidden_size=cfg.v_hidden_size, intermediate_size=cfg.v_intermediate_size, num_hidden_layers=cfg.v_num_hidden_layers,
                     num_attention_heads=cfg.v_num_attention_heads, image_size=cfg.v_image_size, patch_size=cfg.v_patch_size,
                     layer_norm_eps=cfg.v_layer_norm_eps, hidden_act="quick_gelu").save_pretrained(d)
    CLIPImageProcessor(size={"shortest_edge": cfg.v_image_size}, crop_size={"height": cfg.v_image_size, "width": cfg.v_image_size}).save_pretrained(d)
    pre = "vision_model." if layout == "4.31" else ""
    sd = {pre + k[len("vision."):]: torch.from_numpy(v).contiguous() for k, v in wnp.items() if k.startswith("vision.")}
    sd[pre + "post_layernorm.weight"] = torch.ones(cfg.v_hidden_size)       # present in real towers, unused by the path
    sd[pre + "post_layernorm.bias"] = torch.zeros(cfg.v_hidden_size)
    save_file(sd, os.path.join(d, "model.safetensors"))


def write_llava(d, cfg, wnp, clip_dir, with_projector=True):
    from safetensors.torch import save_file
    from synthetic import build as harness
    os.makedirs(d, exist_ok=True)
    lc, _ = harness.hf_configs(cfg)
    lc.mm_vision_tower = clip_dir
    lc.mm_use_im_patch_token = False
    lc.mm_use_im_start_end = False
    lc.architectures = ["LlavaLlamaForCausalLM"]
    lc.save_pretrained(d)
    llm = {}
    for k, v in wnp.items():
        if k.startswith("vision."):
            continue
        if k.startswith("mm_projector."):
            if with_projector:
                llm["model." + k] = torch.from_numpy(v).contiguous()
        else:
            llm[k] = torch.from_numpy(v).contiguous()
    keys = sorted(llm)
    half = len(keys) // 2                                   # two shards, like a real sharded checkpoint
    save_file({k: llm[k] for k in keys[:half]}, os.path.join(d, "model-00001-of-00002.safetensors"))
    save_file({k: llm[k] for k in keys[half:]}, os.path.join(d, "model-00002-of-00002.safetensors"))
    write_tokenizer(d)


