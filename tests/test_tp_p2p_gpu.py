"""One-shot peer-to-peer all-reduce (csrc/p2p.hip) between two PROCESSES: each is one rank of a TP=2 group, both on this
box's single GPU, exchange buffers mapped into each other with HIP IPC (the mechanism used between GPUs), no RCCL in the loop.
Covers the IPC handle exchange, the init self-test and agreement, the flag protocol across processes (1-row decode
all-reduces, a 3-row decode batch, chunked prefill-sized ones), the vocabulary-parallel lm_head whose logits are gathered through the
same kernel (tiny_gqa: V = 320 is not a multiple of H = 256, the partial-last-row form), sampled generation with rank-divergent CPU
generators, and parity with the reference goldens (TP=2 ids == the reference's TP=1 ids)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("dts,name", [("f32", "tiny"), ("bf16", "tiny"), ("f32", "tiny_gqa")])
def test_p2p_allreduce_two_processes(cuda, tmp_path, dts, name):
    world, port = 2, _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"r{r}.json")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_worker.py"), str(r), str(world), str(port), dts, out, name],
                                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), out))
    logs = []
    for p, _ in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
            logs.append("TIMEOUT\n" + o.decode(errors="replace")[-2000:]); continue
        logs.append(o.decode(errors="replace")[-2000:])
    res = []
    for (_, out), lg in zip(procs, logs):
        assert os.path.exists(out), lg
        res.append(json.load(open(out)))
    for r in res:
        assert r["ok"], r.get("trace", r)
        assert r["p2p_active"] and r["status"] == 0
        assert r["big_ok"] and r["status_big"] == 0             # two-shot all-reduce of prefill-sized messages (33 .. 2047 rows, exact data)
        if dts == "f32":
            assert r["logits_err"] <= 1e-3
            assert r["gen"] == r["gen_ref"]
        else:
            assert r["logits_err"] / r["logits_scale"] <= 3e-2
    print({k: v for k, v in res[0].items() if k.startswith("us_per")})
    assert res[0]["gen"] == res[1]["gen"] and res[0]["batch0"] == res[1]["batch0"]
    assert res[0]["vocab_split"]                               # the vocabulary-parallel lm_head + logits gather was on the path
    for r in res:                                              # round 6: the decode batch's RMSNorms ride in the all-reduce launches (one norm launch per step is left)
        assert r["batch_profiled_equal"] and r["batch_linear_launches"] > 0 and 4 * r["batch_rmsnorm_launches"] < r["batch_linear_launches"], r
    assert res[0]["sampled"] == res[1]["sampled"]              # rank 0's sampler seed reached every rank
    if dts == "f32":
        assert res[0]["batch0"][: len(res[0]["gen"][0]) - 1] == res[0]["gen"][0][:-1]     # batch member 0 == the single request


def test_two_shot_allreduce_real_width(cuda, tmp_path):
    """The two-shot all-reduce on messages of the real prefill's size — [1087 .. 1536, 4096] bf16, the engine's exchange region for hidden 4096 (offsets of tens
    of MB, 544 - 768 slices) — between two processes on this GPU: exact sums, no time-out, and the image-feature all-gather of the data-parallel tower
    (model._run_tower).  tests/p2p_big_worker.py says why the sizes stop at 1536 rows and why nothing else may run beside it on a shared GPU."""
    world, port = 2, _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"b{r}.json")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "p2p_big_worker.py"), str(r), str(world), str(port), out],
                                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), out))
    logs = []
    for p, _ in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate()
            logs.append("TIMEOUT\n" + o.decode(errors="replace")[-2000:]); continue
        logs.append(o.decode(errors="replace")[-2000:])
    for (_, out), lg in zip(procs, logs):
        assert os.path.exists(out), lg
        r = json.load(open(out))
        assert r["ok"], r.get("trace", r)
        assert r["p2p_active"] and r["big_ok"] and r["status"] == 0, r
        assert r["gather_ok"] and r["status_gather"] == 0 and r["status_end"] == 0, r
        print({k: round(v, 1) for k, v in r.items() if k.startswith("us_per")})
