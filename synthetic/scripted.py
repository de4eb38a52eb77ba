"""A SCRIPTED model for serving-plumbing workloads (BASELINE config 4): real geometry, real kernels, but weights built
so that greedy decoding emits a fixed text — needed because the LLaVA-Plus tool loop (gradio_web_server_llava_plus.py:498-637) only
turns when the model's answer contains `"thoughts🤔" ... "actions🚀" [...] "value👉" ...`, which random weights never produce.

Construction (workload generator: neither oracle nor product):
  * o_proj and down_proj of every decoder layer are ZERO, so the residual stream at any position is exactly the embedding of the token
    fed there (every GEMM / attention kernel still runs at its real shape and cost);
  * embeddings are random unit-scale rows, `lm_head[j] = embed[j - 1]` for the ids of a scripted chain: after token t the largest logit
    is <rmsnorm(e_t), e_t> at id t + 1 — the model counts upwards along the chain until a row that points at EOS;
  * a table tokenizer spells the scripts with consecutive ids: a prompt ends in a TRIGGER id, chosen by the tokenizer from the prompt
    text (first round: the tool-call script; a prompt that already holds `model outputs:` — the tool answer — the summary script).
Everything between request and response (image encode, splice, prefill, batched decode, streamer, stopping criteria) is the product path."""
from __future__ import annotations

import re
from typing import Dict, List

import numpy as np
import torch

from . import recipes as synth

TOOL_CALL = ('"thoughts🤔" I need the grounding model to find it.\n"actions🚀" [{"API_name": "grounding_dino", "API_params": {"caption": "the object ."}}]\n'
             '"value👉" I will use grounding_dino to help answer.')
SAM_CALL = ('"thoughts🤔" I need a segmentation mask.\n"actions🚀" [{"API_name": "sam", "API_params": {"boxes": [[0.1, 0.2, 0.6, 0.7]]}}]\n'
            '"value👉" I will use sam to help answer.')
SUMMARY = '"thoughts🤔" The tool has answered.\n"actions🚀" []\n"value👉" The object is in the picture, as the tool outputs show.'

BOS, EOS, PAD = 1, 2, 0


def _pieces(text: str) -> List[str]:
    """Split a script into word-ish pieces (each becomes one token id)."""
    return re.findall(r"\s+|[^\s\w]+|\w+", text)


class ScriptedTokenizer:
    """The tokenizer surface the worker path touches (model_worker.py:163-171, mm_utils.py:47-114, TextIteratorStreamer): __call__,
    decode / batch_decode, bos / eos ids, __len__, add_tokens.  Prompt text is byte-hashed into ordinary ids; the LAST id of a prompt is
    the trigger that selects the script the model will recite."""

    def __init__(self, vocab_size: int, scripts: Dict[str, str]):
        self.vocab_size = vocab_size
        self.bos_token_id, self.eos_token_id, self.pad_token_id = BOS, EOS, PAD
        self.bos_token, self.eos_token, self.unk_token = "<s>", "</s>", "<unk>"
        self.id2piece: Dict[int, str] = {}
        self.trigger: Dict[str, int] = {}
        self.chains: Dict[str, List[int]] = {}
        self.base = vocab_size // 2                      # prompt bytes hash into [3, base); the scripts own ids from base upwards
        nxt = self.base
        for name, text in scripts.items():
            self.trigger[name] = nxt
            self.id2piece[nxt] = ""
            ids = []
            for pc in _pieces(text):
                nxt += 1
                self.id2piece[nxt] = pc
                ids.append(nxt)
            self.chains[name] = ids
            nxt += 8
        assert nxt < vocab_size, "vocabulary too small for the scripts"
        self.all_special_ids = [BOS, EOS, PAD]
        self.route = lambda prompt: "summary" if "model outputs:" in prompt else ("sam" if "segment" in prompt.lower() else "tool")

    def __len__(self):
        return self.vocab_size

    def add_tokens(self, toks, special_tokens=False):
        return 0

    def __call__(self, text, **kw):
        class _Enc:
            pass
        e = _Enc()
        raw = text.encode("utf-8")                        # one id per 4 bytes of prompt text
        ids = [BOS] + [3 + (int.from_bytes(raw[i:i + 4], "little") * 2654435761 >> 7) % (self.base - 3) for i in range(0, len(raw), 4)]
        if text.rstrip().endswith("ASSISTANT:"):
            ids.append(self.trigger[self.route(text)])
        e.input_ids = ids
        return e

    def decode(self, ids, skip_special_tokens=False, **kw):
        if torch.is_tensor(ids):
            ids = ids.reshape(-1).tolist()
        out = []
        for i in ids:
            i = int(i)
            if i in (BOS, EOS, PAD):
                if not skip_special_tokens:
                    out.append({BOS: "<s>", EOS: "</s>", PAD: ""}[i])
            else:
                out.append(self.id2piece.get(i, ""))
        return "".join(out)

    def batch_decode(self, seqs, skip_special_tokens=False, **kw):
        return [self.decode(s, skip_special_tokens=skip_special_tokens) for s in seqs]


def scripted_weights_into(model, cfg: synth.SynthConfig, tok: ScriptedTokenizer, seed: int = 0):
    """Load the scripted weights into a product model tensor by tensor (device RNG: a 13B model in seconds)."""
    dev = model.device
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    V, H = cfg.vocab_size, cfg.hidden_size
    embed = torch.randn((V, H), device=dev, generator=gen)
    head = torch.zeros((V, H), device=dev)
    # after the last token of every chain comes EOS: one shared EOS row cannot point at several tokens, so give every chain end the
    # same embedding direction the EOS row points at
    end_vec = torch.randn((H,), device=dev, generator=gen)
    for chain in tok.chains.values():
        embed[chain[-1]] = end_vec
    head[EOS] = end_vec
    for name, chain in tok.chains.items():              # rows that pointed at a chain end must follow its new embedding
        prev = tok.trigger[name]
        for t in chain:
            head[t] = embed[prev]
            prev = t
    for tname, shp in synth.tensor_shapes(cfg).items():
        if tname == "model.embed_tokens.weight":
            t = embed
        elif tname == "lm_head.weight":
            t = head
        elif tname.endswith("self_attn.o_proj.weight") or tname.endswith("mlp.down_proj.weight"):
            t = torch.zeros(shp, device=dev)
        elif tname.endswith("norm.weight") or ("layer_norm" in tname and tname.endswith(".weight")) or tname.endswith("pre_layrnorm.weight"):
            t = torch.ones(shp, device=dev)
        elif tname.endswith(".bias"):
            t = torch.zeros(shp, device=dev)
        else:
            t = torch.randn(shp, device=dev, generator=gen) * 0.02
        model.load_tensor(tname, t)
    model.finalize_weights()
    model.get_vision_tower().is_loaded = True
    return model


def build_scripted(cfg: synth.SynthConfig, dtype=torch.bfloat16, device="cuda", **kw):
    """(tokenizer, model, image_processor, context_len) — the 4-tuple load_pretrained_model returns — for the scripted model."""
    from transformers import CLIPImageProcessor
    from . import build as harness
    from llava_mi355x.model import LlavaLlamaForCausalLM
    tok = ScriptedTokenizer(cfg.vocab_size, {"tool": TOOL_CALL, "sam": SAM_CALL, "summary": SUMMARY})
    lc, vc = harness.hf_configs(cfg)
    lc.eos_token_id, lc.bos_token_id, lc.pad_token_id = EOS, BOS, PAD
    lc.mm_use_im_start_end, lc.mm_use_im_patch_token = False, False
    lc.image_aspect_ratio = "pad"
    lc.max_position_embeddings = 2048
    kw.setdefault("max_position", 2048)
    model = LlavaLlamaForCausalLM(lc, vc, dtype=dtype, device=device, **kw)
    scripted_weights_into(model, cfg, tok)
    S = cfg.v_image_size
    proc = CLIPImageProcessor(size={"shortest_edge": S}, crop_size={"height": S, "width": S})
    model.get_vision_tower().image_processor = proc
    return tok, model, proc, 2048
