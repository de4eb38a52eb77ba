"""Host half of the data-parallel image tower under tensor parallelism (llava_mi355x/model.py: _run_tower, _preencode_requests, the feats_override hand-over; SURVEY §8e
"split images across ranks for B > 1, then all-gather features") with the two engine calls replaced by recorders on CPU tensors: which images a rank encodes, that the
union over the ranks is every image exactly once with zeros elsewhere (so that the all-reduce IS the all-gather), the all-reduce piece sizes, and how the features of a
batch-wide pass reach each request's own encode_images.  The GPU halves: tests/test_tp_gpu.py::test_tp2_tower_data_parallel_over_the_images, tests/test_tp_p2p_gpu.py."""
import ctypes
import threading
import types

import numpy as np
import pytest
import torch


@pytest.fixture()
def M(monkeypatch):
    import llava_mi355x.model as M
    calls = {"encode": [], "allreduce": []}

    def view(p, n, dtype=np.float32):
        addr = p.value if isinstance(p, ctypes.c_void_p) else int(p)
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr))

    class FakeLib:
        @staticmethod
        def lmx_encode_images(h, x, n, out, st):
            P, H = h["P"], h["H"]
            px = view(x, n * 3 * 4 * 4).reshape(n, -1)
            o = view(out, n * P * H).reshape(n, P, H)
            for i in range(n):                                   # "features" = a function of the image's own pixels only
                o[i] = px[i].sum() + np.arange(P * H, dtype=np.float32).reshape(P, H) * 1e-3
            calls["encode"].append(n)
            return 0

        @staticmethod
        def lmx_op_allreduce(h, buf, count, st):
            calls["allreduce"].append(int(count))
            return 0

    monkeypatch.setattr(M, "lib", FakeLib)
    monkeypatch.setattr(M, "stream_handle", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(M, "check", lambda rc, what="": None)
    M._calls = calls
    return M


def _stub(M, rank, world, P=6, H=8):
    s = types.SimpleNamespace()
    s.tp_rank, s.tp_world, s.device, s.dtype = rank, world, torch.device("cpu"), torch.float32
    s.config = types.SimpleNamespace(hidden_size=H)
    s.vision_config = types.SimpleNamespace(image_size=4)
    s.tokens_per_image = P
    s._h = {"P": P, "H": H}
    s._tls = threading.local()
    s._img_cache = None
    s._ensure_final = lambda: None
    for name in ("_check_pixels", "_run_tower", "tower_is_sharded", "encode_images_sharded", "encode_images", "_preencode_requests", "_preencode_local", "_preencode_collective"):
        setattr(s, name, types.MethodType(getattr(M.LlavaLlamaForCausalLM, name), s))
    return s


@pytest.mark.parametrize("world,n", [(2, 2), (2, 5), (4, 3), (8, 8), (8, 19)])
def test_ranks_split_the_images_and_their_buffers_sum_to_the_whole(M, world, n):
    torch.manual_seed(n)
    x = torch.randn(n, 3, 4, 4)
    whole = _stub(M, 0, 1)
    ref = whole.encode_images(x)                                 # unsharded: one engine call over all images
    assert M._calls["encode"] == [n] and M._calls["allreduce"] == []
    total = torch.zeros_like(ref)
    for r in range(world):
        M._calls["encode"].clear(); M._calls["allreduce"].clear()
        m = _stub(M, r, world)
        assert m.tower_is_sharded(n) == (n >= 2)
        out = m.encode_images_sharded(x)
        mine = list(range(r, n, world))
        assert M._calls["encode"] == ([len(mine)] if mine else [])
        for i in range(n):                                       # own images: the unsharded features; every other row: zeros for the sum to fill
            assert torch.equal(out[i], ref[i] if i in mine else torch.zeros_like(ref[i]))
        # the gather goes out in pieces of whole images, <= 4096 rows of H elements each, covering the buffer once
        P, H = 6, 8
        per = max(1, 4096 // P)
        assert M._calls["allreduce"] == [min(per, n - i0) * P * H for i0 in range(0, n, per)]
        total += out
    assert torch.equal(total, ref)


def test_a_request_thread_keeps_the_tower_replicated(M):
    x = torch.randn(3, 3, 4, 4)
    m = _stub(M, 1, 4)
    out = m.encode_images(x)                                     # not the collective entry: every rank encodes everything, no all-reduce
    assert M._calls["encode"] == [3] and M._calls["allreduce"] == []
    assert torch.equal(out, _stub(M, 0, 1).encode_images(x))


def test_batch_wide_pass_hands_each_request_its_rows(M, monkeypatch):
    monkeypatch.delenv("LLAVA_MI355X_TP_TOWER", raising=False)
    m = _stub(M, 0, 2)
    imgs = [torch.randn(1, 3, 4, 4), None, torch.randn(2, 3, 4, 4), [torch.randn(3, 4, 4)]]      # a tensor, no image, two images, a list (left to the request)
    pre = m._preencode_requests(imgs)
    assert pre[1] is None and pre[3] is None
    assert pre[0][0].shape == (1, 6, 8) and pre[2][0].shape == (2, 6, 8)
    assert M._calls["encode"] == [2]                             # rank 0 of 2 took images 0 and 2 of the three
    # the request's own encode_images returns the handed-over rows without touching the engine, once
    M._calls["encode"].clear()
    m._tls.feats_override = pre[2]
    got = m.encode_images(imgs[2])
    assert got is pre[2][0] and M._calls["encode"] == [] and m._tls.feats_override is None
    with pytest.raises(ValueError):
        m._tls.feats_override = pre[0]
        m.encode_images(imgs[2])                                 # rows of another request: refused
    m._tls.feats_override = None
    # the two halves the serving path brackets with agreements: the rank-local one queues no collective, the collective one no tower work
    M._calls["encode"].clear(); M._calls["allreduce"].clear()
    st = m._preencode_local(imgs)
    assert M._calls["encode"] == [2] and M._calls["allreduce"] == []
    pre2 = m._preencode_collective(st)
    assert M._calls["encode"] == [2] and len(M._calls["allreduce"]) == 1
    assert torch.equal(pre2[0][0], pre[0][0]) and torch.equal(pre2[2][0], pre[2][0])
    # switched off: nothing is pre-encoded, every request encodes for itself
    monkeypatch.setenv("LLAVA_MI355X_TP_TOWER", "0")
    assert m._preencode_requests(imgs) == [None] * 4
    # one image in the whole batch: not worth a collective
    monkeypatch.delenv("LLAVA_MI355X_TP_TOWER")
    assert m._preencode_requests([imgs[0], None]) == [None, None]
