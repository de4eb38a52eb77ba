"""Drop-in boundary proof against the UNMODIFIED callers — build container only (needs /root/reference).

A fresh interpreter applies exactly the `sys.modules` aliases INTEGRATION.md §A prescribes, then imports the reference's own, untouched
`llava/serve/model_worker.py` (and `llava/utils.py`, which it pulls in) from /root/reference and checks that
  * every name the worker binds from the model side resolves to THIS package (load_pretrained_model, process_images,
    load_image_from_base64, tokenizer_image_token, KeywordsStoppingCriteria, the constants),
  * its FastAPI routes exist (the worker module really executed to the end),
  * the signatures the callers rely on match the reference's own definitions parameter by parameter: load_pretrained_model
    (llava/model/builder.py:26), LlavaLlamaForCausalLM.forward / prepare_inputs_for_generation (llava_llama.py:56-69, 101),
    encode_images / prepare_inputs_labels_for_multimodal (llava_arch.py:94-101), the mm_utils helpers (mm_utils.py:12-79).
No GPU is needed: nothing is constructed (the product has no CPU path), only imported and inspected."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LLAVA_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "llava", "serve")), reason="reference tree not present (GPU box)")

SCRIPT = r'''
import inspect, json, os, sys, types
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
sys.path.insert(0, ROOT)
out = {}

# ---- the reference's own classes / functions, for the signature comparison (oracle/ref_shim.py: stub packages + exist_ok register)
from oracle import ref_shim
ref = ref_shim.load_reference()
import importlib.util
def ref_module(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m); return m
ref_builder = ref_module("_ref_builder", "llava/model/builder.py")
ref_mm = ref.mm_utils

# ---- INTEGRATION.md §A, verbatim -----------------------------------------------------------------------------------------------
import llava_mi355x.builder, llava_mi355x.mm_utils, llava_mi355x.constants
sys.modules["llava.model.builder"] = llava_mi355x.builder
sys.modules["llava.mm_utils"] = llava_mi355x.mm_utils
sys.modules["llava.constants"] = llava_mi355x.constants
# (the `llava` / `llava.model` package objects are the stubs whose __path__ points into the reference tree, so that
#  `llava.serve.model_worker` and `llava.utils` are the reference's files; importing the reference's llava/__init__.py itself
#  fails under transformers 5.x for reasons unrelated to this path — SURVEY §8b)
sys.modules["llava"].builder = llava_mi355x.builder
for name, rel in (("llava.serve", "llava/serve"),):
    m = types.ModuleType(name); m.__path__ = [os.path.join(REF, rel)]; m.__package__ = name; sys.modules[name] = m

os.chdir(sys.argv[3])                       # the worker's build_logger creates model_worker_<id>.log in cwd
real_stdout = sys.stdout
import llava.serve.model_worker as mw       # the UNMODIFIED file
sys.stdout, sys.stderr = real_stdout, sys.__stderr__     # build_logger redirected them into the log

out["worker_file"] = mw.__file__
out["utils_file"] = sys.modules["llava.utils"].__file__
import llava_mi355x.builder as B, llava_mi355x.mm_utils as M, llava_mi355x.constants as C
out["resolves"] = {
    "load_pretrained_model": mw.load_pretrained_model is B.load_pretrained_model,
    "process_images": mw.process_images is M.process_images,
    "load_image_from_base64": mw.load_image_from_base64 is M.load_image_from_base64,
    "tokenizer_image_token": mw.tokenizer_image_token is M.tokenizer_image_token,
    "KeywordsStoppingCriteria": mw.KeywordsStoppingCriteria is M.KeywordsStoppingCriteria,
    "IMAGE_TOKEN_INDEX": mw.IMAGE_TOKEN_INDEX == C.IMAGE_TOKEN_INDEX == -200,
    "DEFAULT_IMAGE_TOKEN": mw.DEFAULT_IMAGE_TOKEN == "<image>",
    "WORKER_HEART_BEAT_INTERVAL": mw.WORKER_HEART_BEAT_INTERVAL == C.WORKER_HEART_BEAT_INTERVAL,
}
out["routes"] = sorted(r.path for r in mw.app.routes if hasattr(r, "path"))
out["has_ModelWorker"] = inspect.isclass(mw.ModelWorker) and hasattr(mw.ModelWorker, "generate_stream")

def params(fn):
    return [(p.name, None if p.default is inspect._empty else repr(p.default), str(p.kind)) for p in inspect.signature(fn).parameters.values()]

from llava_mi355x.model import LlavaLlamaForCausalLM as Ours
Ref = ref.LlavaLlamaForCausalLM
out["sig"] = {
    "load_pretrained_model": (params(ref_builder.load_pretrained_model), params(B.load_pretrained_model)),
    "forward": (params(Ref.forward), params(Ours.forward)),
    "prepare_inputs_for_generation": (params(Ref.prepare_inputs_for_generation), params(Ours.prepare_inputs_for_generation)),
    "encode_images": (params(Ref.encode_images), params(Ours.encode_images)),
    "prepare_inputs_labels_for_multimodal": (params(Ref.prepare_inputs_labels_for_multimodal), params(Ours.prepare_inputs_labels_for_multimodal)),
    "process_images": (params(ref_mm.process_images), params(M.process_images)),
    "tokenizer_image_token": (params(ref_mm.tokenizer_image_token), params(M.tokenizer_image_token)),
    "expand2square": (params(ref_mm.expand2square), params(M.expand2square)),
    "load_image_from_base64": (params(ref_mm.load_image_from_base64), params(M.load_image_from_base64)),
    "get_model_name_from_path": (params(ref_mm.get_model_name_from_path), params(M.get_model_name_from_path)),
    "KeywordsStoppingCriteria.__init__": (params(ref_mm.KeywordsStoppingCriteria.__init__), params(M.KeywordsStoppingCriteria.__init__)),
    "KeywordsStoppingCriteria.__call__": (params(ref_mm.KeywordsStoppingCriteria.__call__), params(M.KeywordsStoppingCriteria.__call__)),
}
# attributes the worker / builder / CLI touch on the model object (model_worker.py:136-158, builder.py:132-144, cli.py:56-75)
out["surface"] = {a: hasattr(Ours, a) for a in ("generate", "forward", "get_vision_tower", "get_model", "resize_token_embeddings", "from_pretrained",
                                                 "encode_images", "prepare_inputs_labels_for_multimodal", "prepare_inputs_for_generation", "to", "eval")}
print("RESULT" + json.dumps(out))
'''


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    work = tmp_path_factory.mktemp("worker_cwd")
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF, str(work)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and lines, f"probe failed (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return json.loads(lines[-1][len("RESULT"):])


def test_unmodified_worker_imports_on_this_package(probe):
    assert os.path.realpath(probe["worker_file"]) == os.path.realpath(os.path.join(REF, "llava/serve/model_worker.py"))
    assert os.path.realpath(probe["utils_file"]) == os.path.realpath(os.path.join(REF, "llava/utils.py"))
    bad = [k for k, ok in probe["resolves"].items() if not ok]
    assert not bad, f"the worker did not bind these names to this package: {bad}"
    assert "/worker_generate_stream" in probe["routes"] and "/worker_get_status" in probe["routes"]
    assert probe["has_ModelWorker"]
    missing = [k for k, ok in probe["surface"].items() if not ok]
    assert not missing, f"model surface the callers use is missing: {missing}"


def _names(ps):
    return [p[0] for p in ps]


@pytest.mark.parametrize("fn", ["load_pretrained_model", "forward", "prepare_inputs_for_generation", "encode_images",
                                "prepare_inputs_labels_for_multimodal", "process_images", "tokenizer_image_token", "expand2square",
                                "load_image_from_base64", "get_model_name_from_path", "KeywordsStoppingCriteria.__init__",
                                "KeywordsStoppingCriteria.__call__"])
def test_signature_matches_reference(probe, fn):
    """Every parameter of the reference, in the reference's order, with the reference's default; this build may only APPEND optional
    keyword parameters (torch_dtype / tp_rank / ... on load_pretrained_model, **kwargs on forward)."""
    ref, ours = probe["sig"][fn]
    ref = [p for p in ref if "VAR_" not in p[2]]
    ours_fixed = [p for p in ours if "VAR_" not in p[2]]
    assert _names(ours_fixed)[: len(ref)] == _names(ref), f"{fn}: reference {_names(ref)} vs this build {_names(ours_fixed)}"
    for r, o in zip(ref, ours_fixed):
        assert r[1] == o[1], f"{fn}: default of `{r[0]}` is {o[1]} here, {r[1]} in the reference"
    for extra in ours_fixed[len(ref):]:
        assert extra[1] is not None, f"{fn}: extra parameter `{extra[0]}` has no default — existing callers would break"
