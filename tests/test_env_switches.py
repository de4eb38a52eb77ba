"""Guard for the prune of round 5: the library keeps at most 12 `LMX_*` environment switches, and none of them is read on a launch path —
every `getenv` sits in a function-local `static` initialiser (read once per process), in `Model::Model` (read once per model) or in the two set-up functions of the
tensor-parallel group / the decode batch's weight copy.  A new experiment arm has to bring its measurement, not a switch."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llava-plus-codebase_amd", "csrc")
SETUP_FUNCTIONS = ("Model::Model(", "void Model::p2p_connect(", "void Model::ensure_batch_weights(")


def _sources():
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".h")):
            yield f, open(os.path.join(CSRC, f)).read()


def test_at_most_twelve_switches_and_none_on_a_launch_path():
    names, bad = set(), []
    for f, text in _sources():
        lines = text.split("\n")
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            for m in re.finditer(r'getenv\("([A-Z0-9_]+)"\)', code):
                names.add(m.group(1))
                if re.search(r"static const \w+ \w+ = \[\]", code):
                    continue                                   # function-local static: evaluated once
                # otherwise the enclosing function must be one of the set-up functions: walk back to the nearest line that opens a function at column 0
                j = i
                while j >= 0 and not re.match(r"^[A-Za-z_].*\)\s*(const\s*)?(:.*)?\{\s*$", lines[j]):
                    j -= 1
                head = lines[j] if j >= 0 else ""
                if not any(h in head for h in SETUP_FUNCTIONS):
                    bad.append((f, i + 1, m.group(1), head.strip()[:80]))
    assert not bad, bad
    assert all(n.startswith("LMX_") for n in names), names
    assert len(names) <= 12, sorted(names)


def test_removed_arms_stay_removed():
    """kernels and switches the round-5 prune deleted (profiles/EXPERIMENTS.md r5-D) do not come back by accident"""
    gone = ["decode_flow_kernel", "decode_attn_head_kernel", "decode_attn_o_kernel", "gemv2m_kernel", "prefetch_kernel", "splitk_reduce_norm_kernel",
            "splitk_reduce_rows_norm_kernel", "splitk_reduce_rowmajor_kernel", "splitk_reduce_hyb_kernel", "LMX_DECODE_FLOW", "LMX_ATTN_MERGE", "LMX_FUSED_AO",
            "LMX_SPLITK_MODE", "LMX_GEMM8P_TAIL", "LMX_GEMM8P_MTAIL", "LMX_FUSE_NORM", "LMX_SKINNY_XNORM", "LMX_DECODE_PREFETCH", "LMX_ATTN_HEAD", "LMX_ATTN_WAVE", "LMX_BATCH_ATTN"]
    hits = []
    for f, text in _sources():
        code = "\n".join(l.split("//")[0] for l in text.split("\n"))
        hits += [(f, g) for g in gone if g in code]
    assert not hits, hits
