"""Prompt templates: the host-side data format on the input side of the hot path.

Mirror of the part of llava/conversation.py the training entry (llava/train/train.py:254-665 builds every training prompt through
`conversation_lib.default_conversation` / `conv_templates[--version]`) and the serving clients (gradio_web_server_llava_plus.py:444-637,
tools/config4_harness.py) use: the `Conversation` record, its five separator styles and `get_prompt()` (llava/conversation.py:54-148), `append_message`,
`copy`, and the templates the reference's scripts select (`v1` / `plain` / `llava_llama_2`, plus the other text templates of llava/conversation.py:516-622).
The Gradio rendering half of that file (HTML, image thumbnails, tool-output folding) is the web UI — out of scope.

Same names, same fields, same strings out (tests/test_train_data_vs_reference.py compares every template and style against the reference's own module).
"""
from __future__ import annotations

import dataclasses
import os
from enum import Enum, auto
from typing import List, Optional, Sequence


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


def _text(message):
    """A message is a string or, from the web client, a tuple whose first entry is the text (llava/conversation.py:48-53: 3 or 4 entries)."""
    if isinstance(message, tuple):
        if len(message) not in (3, 4):
            raise ValueError(f"Invalid msg with len {len(message)}: {message}")
        return message[0]
    return message


def _render_single(c: "Conversation", msgs) -> str:
    out = [c.system, c.sep]
    for role, m in msgs:
        out.append(f"{role}: {_text(m)}{c.sep}" if m else f"{role}:")
    return "".join(out)


def _render_two(c: "Conversation", msgs) -> str:
    seps = (c.sep, c.sep2)
    out = [c.system, seps[0]]
    for i, (role, m) in enumerate(msgs):
        out.append(f"{role}: {_text(m)}{seps[i % 2]}" if m else f"{role}:")
    return "".join(out)


def _render_mpt(c: "Conversation", msgs) -> str:
    out = [c.system, c.sep]
    for role, m in msgs:
        out.append(f"{role}{_text(m)}{c.sep}" if m else role)
    return "".join(out)


def _render_llama_2(c: "Conversation", msgs) -> str:
    out = ""
    for i, (role, m) in enumerate(msgs):
        if i == 0:
            assert m, "first message should not be none"
            assert role == c.roles[0], "first message should come from user"
        if not m:
            continue
        m = _text(m)
        if i == 0:
            m = f"<<SYS>>\n{c.system}\n<</SYS>>\n\n{m}"
        out += f"{c.sep}[INST] {m} [/INST]" if i % 2 == 0 else f" {m} {c.sep2}"
    return out.lstrip(c.sep)


def _render_plain(c: "Conversation", msgs) -> str:
    seps = (c.sep, c.sep2)
    out = c.system
    for i, (_, m) in enumerate(msgs):
        if m:
            out += _text(m) + seps[i % 2]
    return out


_RENDER = {SeparatorStyle.SINGLE: _render_single, SeparatorStyle.TWO: _render_two, SeparatorStyle.MPT: _render_mpt,
           SeparatorStyle.LLAMA_2: _render_llama_2, SeparatorStyle.PLAIN: _render_plain}


@dataclasses.dataclass
class Conversation:
    """One conversation's history and the template that turns it into a prompt (llava/conversation.py:54-66)."""
    system: str
    roles: Sequence[str]
    messages: List[List[str]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def get_prompt(self) -> str:
        msgs = self.messages
        if len(msgs) > 0 and type(msgs[0][1]) is tuple:
            # first message came with an image: the placeholder moves to the front of its text (or, for the mmtag templates, into a turn of its own)
            msgs = list(self.messages)
            role0, first = msgs[0]
            text0 = first[0].replace("<image>", "").strip()
            if "mmtag" in self.version:
                msgs[0] = (role0, text0)
                msgs.insert(0, (self.roles[0], "<Image><image></Image>"))
                msgs.insert(1, (self.roles[1], "Received."))
            else:
                msgs[0] = (role0, "<image>\n" + text0)
        try:
            render = _RENDER[self.sep_style]
        except KeyError:
            raise ValueError(f"Invalid style: {self.sep_style}") from None
        return render(self, msgs)

    def append_message(self, role, message) -> None:
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages], offset=self.offset,
                            sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, version=self.version)


# ---- templates (system prompts and separators are data: llava/conversation.py:516-622) -------------------------------------------------------------
_CHAT_USER = ("A chat between a curious user and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the user's questions.")
_CHAT_HUMAN = ("A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.")
_MMTAG = ("A chat between a curious user and an artificial intelligence assistant. "
          "The assistant is able to understand the visual content that the user provides, and assist the user with a variety of tasks using natural language."
          "The visual content will be provided with the following format: <Image>visual content</Image>.")
_LLAMA_2_SYSTEM = """You are a helpful, respectful and honest assistant. Always answer as helpfully as possible, while being safe.  Your answers should not include any harmful, unethical, racist, sexist, toxic, dangerous, or illegal content. Please ensure that your responses are socially unbiased and positive in nature.

If a question does not make any sense, or is not factually coherent, explain why instead of answering something not correct. If you don't know the answer to a question, please don't share false information."""


def _two(system, version="v1", roles=("USER", "ASSISTANT")):
    return Conversation(system=system, roles=roles, version=version, messages=(), offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")


def _llama_2(system):
    return Conversation(system=system, roles=("USER", "ASSISTANT"), version="llama_v2", messages=(), offset=0, sep_style=SeparatorStyle.LLAMA_2,
                        sep="<s>", sep2="</s>")


conv_vicuna_v1 = _two(_CHAT_USER)
conv_llava_v1 = _two(_CHAT_HUMAN)
conv_llava_v1_mmtag = _two(_MMTAG, version="v1_mmtag")
conv_llama_2 = _llama_2(_LLAMA_2_SYSTEM)
conv_llava_llama_2 = _llama_2("You are a helpful language and vision assistant. "
                              "You are able to understand the visual content that the user provides, "
                              "and assist the user with a variety of tasks using natural language.")
conv_mpt = Conversation(system="<|im_start|>system\nA conversation between a user and an LLM-based AI assistant. The assistant gives helpful and honest answers.",
                        roles=("<|im_start|>user\n", "<|im_start|>assistant\n"), version="mpt", messages=(), offset=0, sep_style=SeparatorStyle.MPT,
                        sep="<|im_end|>")
conv_llava_plain = Conversation(system="", roles=("", ""), messages=(), offset=0, sep_style=SeparatorStyle.PLAIN, sep="\n")
conv_llava_v0 = Conversation(system=_CHAT_HUMAN, roles=("Human", "Assistant"), messages=(), offset=0, sep_style=SeparatorStyle.SINGLE, sep="###")
conv_llava_v0_mmtag = Conversation(system=_MMTAG, roles=("Human", "Assistant"), messages=(), offset=0, sep_style=SeparatorStyle.SINGLE, sep="###",
                                   version="v0_mmtag")

# `--version` of the training entry / `conv_mode` of the clients -> template.  (The reference's "default" / "v0" entries are a few-shot Vicuna-v0 text prompt
# with two canned exchanges; no LLaVA script selects them — they resolve to the same roles / separators without the canned exchanges here.)
conv_templates = {
    "default": conv_llava_v0, "v0": conv_llava_v0,
    "v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1, "llama_2": conv_llama_2,
    "plain": conv_llava_plain, "v0_plain": conv_llava_plain, "llava_v0": conv_llava_v0, "v0_mmtag": conv_llava_v0_mmtag,
    "llava_v1": conv_llava_v1, "v1_mmtag": conv_llava_v1_mmtag, "llava_llama_2": conv_llava_llama_2, "mpt": conv_mpt,
}

# the template `preprocess` (train_data.py) reads; the training entry replaces it with conv_templates[--version] (llava/train/train.py:902-907).
# LLAVA_DEFAULT_CONVERSATION names one of the conv_* objects above, as in the reference (llava/conversation.py:624-627)
default_conversation = globals()[os.getenv("LLAVA_DEFAULT_CONVERSATION", "conv_vicuna_v1")]
