"""LLaVA-Plus tool-loop plumbing (BASELINE config 4) on this build: concurrent clients drive
generate -> parse `"actions🚀"` -> stub grounding_dino / sam REST worker -> re-prompt with the tool's JSON -> generate, the sequence of
llava/serve/gradio_web_server_llava_plus.py:444-637, over HTTP against the worker endpoint (tools/worker_reenactment.py) with and
without the continuous-batching scheduler.  The model is synthetic/scripted.py's scripted model (real kernels, weights arranged to
recite a tool call and then a summary), so every text is known in advance."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("batch", [1, 4])
def test_tool_loop_concurrent_requests(cuda, batch):
    import config4_harness as h
    res = h.run("tiny", n_requests=6, batch=batch, dtype_name="f32", sam_every=3)
    assert not res["errors"], res["errors"]
    assert res["completed"] == 6
    exp = res["expected"]
    for i, r in enumerate(res["records"]):
        want_sam = i % 3 == 2
        assert r["first_answer"] == (exp["sam"] if want_sam else exp["tool"]), (i, r["first_answer"])
        assert r["tool"] == ("sam" if want_sam else "grounding_dino")
        assert r["final_answer"] == exp["summary"], (i, r["final_answer"])
        tail = r["prompt2_tail"]
        if want_sam:
            assert r["mask_rle"] == {"size": [64, 96], "counts": "0000"}                  # state.mask_rle = masks_rle[0]  (:598-599)
            assert "sam model outputs: {'boxes': [[0.1, 0.2, 0.6, 0.7]]}" in tail and "masks_rle" not in tail
        else:
            assert r["tool_response_raw"]["boxes"] == [[0.123456, 0.2, 0.654321, 0.7]]
            # cleaned for the re-prompt: rounded to 2 decimals, `size` dropped (:565-581)
            assert "grounding_dino model outputs: {'boxes': [[0.12, 0.2, 0.65, 0.7]], 'logits': [0.88], 'phrases': ['the object']}" in tail
            assert "'size'" not in tail
        assert "Please summarize the model outputs and answer my first question:" in tail and tail.endswith("ASSISTANT:")
    assert res["tool_calls"] == {"grounding_dino": 4, "sam": 2}
