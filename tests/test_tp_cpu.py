"""Multi-process CPU test (gloo, world_size 2) of the tensor-parallel decoder algebra the engine implements:
each rank holds the shard `tests/tp_shards.py: shard_tensor` selects (the same slices Model::load_weight copies), runs its
part of every decoder layer, and the partial o_proj / down_proj outputs are all-reduced (residual on rank 0 only); the lm_head is
vocabulary-parallel, its logits gathered by a sum over zero-padded rows — the result must equal the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, name, out_path):
    for p in (os.path.join(ROOT, "llava-plus-codebase_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import torch.nn.functional as F
    from tp_shards import shard_tensor
    from oracle import llava_oracle as O
    from synthetic import recipes as synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.CONFIGS[name]
    w = O.to_torch_weights(synth.make_weights(cfg, 0))
    nh, nkv, d, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size
    sh = {k: shard_tensor(k, v, nh, nkv, d, I, rank, world) for k, v in w.items()}
    nh_l, nkv_l = nh // world, nkv // world
    torch.manual_seed(0)
    T = 9
    h = torch.randn(1, T, cfg.hidden_size)
    pos = torch.arange(T)[None]
    cos, sin = O.rope_cos_sin(cfg, pos, torch.float32)
    bias = torch.zeros(1, 1, T, T).masked_fill(~(torch.arange(T)[None, :] <= torch.arange(T)[:, None])[None, None], float("-inf"))
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        x = O.rms_norm(h, sh[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(x, sh[p + "self_attn.q_proj.weight"]).view(1, T, nh_l, d).transpose(1, 2)
        k = F.linear(x, sh[p + "self_attn.k_proj.weight"]).view(1, T, nkv_l, d).transpose(1, 2)
        v = F.linear(x, sh[p + "self_attn.v_proj.weight"]).view(1, T, nkv_l, d).transpose(1, 2)
        q = q * cos[:, None] + O.rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + O.rotate_half(k) * sin[:, None]
        kk = k.repeat_interleave(nh_l // nkv_l, dim=1); vv = v.repeat_interleave(nh_l // nkv_l, dim=1)
        a = torch.softmax(q @ kk.transpose(2, 3) / d ** 0.5 + bias, dim=-1)
        o = (a @ vv).transpose(1, 2).reshape(1, T, nh_l * d)
        part = F.linear(o, sh[p + "self_attn.o_proj.weight"])
        if rank == 0:
            part = part + h                      # residual on rank 0's partial only (engine: `lead ? h : nullptr`)
        dist.all_reduce(part)
        h = part
        x = O.rms_norm(h, sh[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        act = F.silu(F.linear(x, sh[p + "mlp.gate_proj.weight"])) * F.linear(x, sh[p + "mlp.up_proj.weight"])
        part = F.linear(act, sh[p + "mlp.down_proj.weight"])
        if rank == 0:
            part = part + h
        dist.all_reduce(part)
        h = part
    # vocabulary-parallel lm_head: each rank fills its column block of a ZEROED logits row, the sum over ranks is the all-gather
    # (Model::gather_logits) — exact, every other rank contributes zeros
    xl = O.rms_norm(h[:, -1:], sh["model.norm.weight"], cfg.rms_norm_eps)
    V = cfg.vocab_size
    logits = torch.zeros(1, 1, V)
    wl = sh["lm_head.weight"]
    v_l = wl.shape[0]
    off = rank * v_l if v_l != V else 0
    logits[..., off:off + v_l] = F.linear(xl, wl)
    if v_l != V:
        dist.all_reduce(logits)
    if rank == 0:
        torch.manual_seed(0)
        h0 = torch.randn(1, T, cfg.hidden_size)
        hr = h0
        for i in range(cfg.num_hidden_layers):
            hr, _ = O.decoder_layer(w, cfg, i, hr, cos, sin, None, bias)
        ref_logits = F.linear(O.rms_norm(hr[:, -1:], w["model.norm.weight"], cfg.rms_norm_eps), w["lm_head.weight"])
        np.save(out_path, np.array([(h - hr).abs().max().item(), hr.abs().max().item(), (logits - ref_logits).abs().max().item(),
                                    ref_logits.abs().max().item(), float(v_l != V)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny", "tiny_gqa"])
def test_tp2_matches_unsharded(tmp_path, name):
    out = str(tmp_path / "err.npy")
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    err, mag, lerr, lmag, split = np.load(out)
    assert err <= 1e-4 * max(mag, 1.0), (err, mag)
    assert split == 1.0 and lerr <= 1e-4 * max(lmag, 1.0), (lerr, lmag, split)


def test_shard_slices_cover_and_partition():
    from tp_shards import shard_slices
    nh, nkv, d, I, H = 8, 4, 64, 1024, 512
    for world in (2, 4):
        for name, shape in (("model.layers.0.self_attn.q_proj.weight", (nh * d, H)), ("model.layers.0.self_attn.k_proj.weight", (nkv * d, H)),
                            ("model.layers.3.self_attn.o_proj.weight", (H, nh * d)), ("model.layers.1.mlp.gate_proj.weight", (I, H)),
                            ("model.layers.1.mlp.down_proj.weight", (H, I))):
            seen = np.zeros(shape, int)
            for r in range(world):
                rs, cs = shard_slices(name, shape, nh, nkv, d, I, r, world)
                seen[rs, cs] += 1
            assert (seen == 1).all(), name
        assert shard_slices("model.norm.weight", (H,), nh, nkv, d, I, 0, world) is None
        seen = np.zeros((32000, 8), int)
        for r in range(world):
            rs, cs = shard_slices("lm_head.weight", (32000, 8), nh, nkv, d, I, r, world)
            seen[rs, cs] += 1
        assert (seen == 1).all()
        assert shard_slices("lm_head.weight", (32001, 8), nh, nkv, d, I, 0, world) is None      # does not split evenly: replicated
        assert shard_slices("lm_head.weight", (100, H), nh, nkv, d, I, 1, world) is None


def _channel_worker(rank, world, port, out_path):
    """Leader/follower command log over gloo (tp_serving.CommandChannel): rank 0 announces a serving session, rank 1 records it."""
    import json
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
    from llava_mi355x.tp_serving import CommandChannel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    chan = CommandChannel(dist.new_group(backend="gloo"))
    if rank == 0:
        req = {"ids": torch.arange(7)[None], "images": [torch.ones(3, 4, 4, dtype=torch.float16)], "attention_mask": None,
               "sampling": (0.8, 0.9, None, 123456789012345), "prefill_chunk": 0}
        chan.send(("prefill", 0, CommandChannel.wire_request(req)))
        chan.send(("step", [0]))
        chan.send(("prefill", 1, CommandChannel.wire_request(dict(req, sampling=None))))
        for _ in range(3):
            chan.send(("step", [0, 1]))
        chan.send(("release", 0)); chan.send(("step", [1])); chan.send(("release", 1)); chan.send(("stop",))
        assert chan.sent == 10
    else:
        log = []
        while True:
            cmd = chan.recv()
            if cmd[0] == "prefill":
                r = cmd[2]
                assert torch.equal(r["ids"], torch.arange(7)[None]) and r["images"][0].dtype == torch.float16 and r["images"][0].shape == (3, 4, 4)
                log.append(["prefill", cmd[1], r["sampling"][3] if r["sampling"] else None])
            else:
                log.append(list(cmd))
            if cmd[0] == "stop":
                break
        json.dump(log, open(out_path, "w"))
    dist.barrier(); dist.destroy_process_group()


def test_command_channel_order_and_payload(tmp_path):
    import json
    out = str(tmp_path / "log.json")
    mp.spawn(_channel_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    log = json.load(open(out))
    assert log == [["prefill", 0, 123456789012345], ["step", [0]], ["prefill", 1, None], ["step", [0, 1]], ["step", [0, 1]], ["step", [0, 1]],
                   ["release", 0], ["step", [1]], ["release", 1], ["stop"]]


def _symmetric_worker(rank, world, port, out_path):
    """tp_serving.prefill_symmetric over gloo with a stand-in model: a request that ONE rank cannot prepare is dropped on every rank before the
    collective-bearing half runs, a failure inside that half fails the same requests everywhere, and the step status exchange reports a rank that
    could not launch (a one-rank failure used to leave the other ranks blocked in RCCL)."""
    import json
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
    from llava_mi355x.tp_serving import CommandChannel, prefill_symmetric
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    chan = CommandChannel(dist.new_group(backend="gloo"))

    class Cache:
        def __init__(self, tag): self.tag, self.closed = tag, False
        def close(self): self.closed = True

    class Model:
        device = torch.device("cpu")
        def __init__(self, bad_prepare, bad_run): self.bad_prepare, self.bad_run, self.ran, self.made = bad_prepare, bad_run, [], []
        def _prepare_request(self, ids, images, attention_mask, sampling, stop=None):
            tag = int(ids[0, 0])
            if tag in self.bad_prepare:
                raise MemoryError(f"rank {rank} cannot allocate request {tag}")
            c = Cache(tag); self.made.append(c)
            return {"cache": c, "embeds": None, "valid": None}
        def _run_prepared(self, prepared, chunk, return_logits=False):
            self.ran.append([p["cache"].tag for p in prepared])
            if self.bad_run:
                raise RuntimeError(f"rank {rank}: packed prefill failed")

    reqs = [{"ids": torch.tensor([[t, 1, 2]]), "images": None, "attention_mask": None, "sampling": None} for t in (10, 11, 12)]
    log = {}
    # 1) request 11 cannot be prepared on rank 1 only
    m = Model({11} if rank == 1 else set(), False)
    res = prefill_symmetric(m, chan, reqs, 0)
    log["drop"] = {"kinds": [type(r).__name__ for r in res], "ran": m.ran, "closed": [c.tag for c in m.made if c.closed]}
    # 2) the collective-bearing half fails on rank 0 only: everything fails everywhere, every sequence is released
    m = Model(set(), rank == 0)
    res = prefill_symmetric(m, chan, reqs, 0)
    log["run"] = {"kinds": [isinstance(r, BaseException) for r in res], "closed": sorted(c.tag for c in m.made if c.closed)}
    # 3) step status: rank 1 reports a failed launch after the second step
    st = [chan.agree_end(chan.agree_begin(True)), chan.agree_end(chan.agree_begin(rank == 0)), chan.agree_end(None)]
    log["steps"] = st

    # 4) sharded image encode (step 0 of prefill_symmetric): rank 1 does not hold request 11's pixels -> the batch-wide tower pass runs over requests 10 and 12
    #    on BOTH ranks and request 11 takes its own (replicated) encode
    class TPModel(Model):
        tp_world = 2
        def __init__(self, bad_encode): super().__init__(set(), False); self.bad_encode, self.enc, self.got, self.coll, self.cleared = bad_encode, [], [], 0, 0
        def _check_pixels(self, im):
            if im.shape[1:] != (3, 4, 4): raise ValueError("geometry")
            return im
        def tower_is_sharded(self, n): return n >= 2
        def _preencode_local(self, images):
            self.enc.append([None if im is None else int(im.shape[0]) for im in images])
            if self.bad_encode == "local": raise MemoryError(f"rank {rank}: cannot allocate the tower pass")
            return {"images": images}
        def _preencode_collective(self, st):
            self.coll += 1
            if self.bad_encode == "collective": raise RuntimeError(f"rank {rank}: tower all-gather failed")
            return [None if im is None else ("feats", int(im.shape[0])) for im in st["images"]]
        def clear_image_cache(self): self.cleared += 1
        def _prepare_request(self, ids, images, attention_mask, sampling, stop=None, feats=None):
            self.got.append(feats)
            return super()._prepare_request(ids, images, attention_mask, sampling, stop)
    img = lambda k: torch.zeros(k, 3, 4, 4)
    reqs_i = [dict(r, images=im) for r, im in zip(reqs, (img(1), img(2) if rank == 0 else None, img(1)))]
    m = TPModel(False)
    res = prefill_symmetric(m, chan, reqs_i, 0)
    log["shard"] = {"enc": m.enc, "got": m.got, "kinds": [type(r).__name__ for r in res]}
    # 5) the RANK-LOCAL half of the pass fails on rank 0 only: agreed on before any rank queued a collective -> every request fails on every rank, serving goes on
    m = TPModel("local" if rank == 0 else None)
    res = prefill_symmetric(m, chan, [dict(r, images=img(1)) for r in reqs], 0)
    log["shard_fail"] = {"kinds": [isinstance(r, BaseException) for r in res], "made": len(m.made), "coll": m.coll}
    # 6) the COLLECTIVE half fails on rank 0 only: some stream now holds a collective the others do not match -> every rank gives the group up
    from llava_mi355x.tp_serving import TensorParallelDesync
    m = TPModel("collective" if rank == 0 else None)
    try:
        prefill_symmetric(m, chan, [dict(r, images=img(1)) for r in reqs], 0)
        log["shard_desync"] = {"raised": False}
    except TensorParallelDesync:
        log["shard_desync"] = {"raised": True, "cleared": m.cleared, "made": len(m.made)}
    json.dump(log, open(f"{out_path}.{rank}", "w"))
    dist.barrier(); dist.destroy_process_group()


def test_tensor_parallel_failures_are_symmetric(tmp_path):
    import json
    out = str(tmp_path / "sym.json")
    mp.spawn(_symmetric_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = json.load(open(out + ".0")), json.load(open(out + ".1"))
    for r in (r0, r1):
        assert r["drop"]["ran"] == [[10, 12]]                                  # the same packed call on both ranks, without the request rank 1 lost
        assert r["run"]["kinds"] == [True, True, True] and r["run"]["closed"] == [10, 11, 12]
        assert r["steps"] == [True, False, True]
        assert r["shard"]["enc"] == [[1, None, 1]] and r["shard"]["got"] == [["feats", 1], None, ["feats", 1]] and r["shard"]["kinds"] == ["Cache"] * 3
        assert r["shard_fail"] == {"kinds": [True, True, True], "made": 0, "coll": 0}     # nobody queued the collective or went on to allocate sequences
        assert r["shard_desync"] == {"raised": True, "cleared": 1, "made": 0}
    assert r0["drop"]["kinds"] == ["Cache", "RuntimeError", "Cache"] and r0["drop"]["closed"] == [11]       # rank 0 had prepared it: released
    assert r1["drop"]["kinds"] == ["Cache", "MemoryError", "Cache"] and r1["drop"]["closed"] == []


def _zero_worker(rank, world, port, out_path):
    """ZeRO-2 bookkeeping over gloo (llava_mi355x/train.py ZeroPartition): bucket reduce-scatter (sum) and all-gather on host tensors."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))
    from llava_mi355x.train import ZeroPartition
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sizes = [("lm_head", 5000), ("norm", 64), ("l1.down", 9000), ("l1.qkv", 3 * 4096), ("l0.down", 9000), ("embed", 5000)]
    P = ZeroPartition(sizes, world, rank, bucket_elems=8192)
    gen = torch.Generator().manual_seed(100 + rank)
    flat_g = torch.randn(P.total, generator=gen)
    g_shard = torch.zeros(P.shard_elems); p_shard = torch.zeros(P.shard_elems)
    for b in range(len(P.buckets)):
        P.reduce_scatter(flat_g, g_shard, b, None)
    # what every rank's flat gradient was (same generator seeds), summed
    total = sum(torch.randn(P.total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    ok = True
    for b in range(len(P.buckets)):
        o0, o1 = P.owned(b)
        ok &= bool(torch.allclose(P.shard_view(g_shard, b), total[o0:o1], atol=1e-6))
        P.shard_view(p_shard, b).copy_(total[o0:o1] * 0.5)               # "updated parameters" of the owned slice
    flat_p = torch.zeros(P.total)
    for b in range(len(P.buckets)):
        P.all_gather(flat_p, p_shard, b, None)
    ok &= bool(torch.allclose(flat_p, total * 0.5, atol=1e-6))
    x = torch.tensor([float(rank + 1)]); P.all_reduce_scalar(x); ok &= x.item() == world * (world + 1) / 2
    if rank == 0:
        import json
        json.dump({"ok": ok, "buckets": P.buckets, "members": P.members, "total": P.total, "shard": P.shard_elems}, open(out_path, "w"))
    dist.barrier(); dist.destroy_process_group()


def test_zero2_partition_collectives_gloo(tmp_path):
    import json
    out = str(tmp_path / "zero.json")
    mp.spawn(_zero_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = json.load(open(out))
    assert r["ok"]
    assert len(r["buckets"]) >= 3 and r["shard"] * 2 == r["total"]
    assert all((e - s) % (2 * 64) == 0 for s, e in r["buckets"])                       # every bucket splits evenly, slices stay 64-aligned
    assert [n for m in r["members"] for n in m] == ["lm_head", "norm", "l1.down", "l1.qkv", "l0.down", "embed"]      # backward order kept


@pytest.mark.parametrize("world", [1, 2, 3, 5, 6, 7, 8])
def test_zero_partition_keeps_qkv_adjacent_for_any_world(world):
    """with the default bucket (hidden^2 elements) a bucket used to close right after q_proj and its padding (a multiple of world x 64
    elements) separated k_proj from q_proj for world = 3, 5, 6, 7 — TrainStep's fused q|k|v operand then failed its adjacency assertion.  The
    reference's ZeRO-2 accepts any world size (scripts/zero2.json)."""
    sys.path.insert(0, os.path.join(ROOT, "llava-plus-codebase_amd"))
    from llava_mi355x.train import ZeroPartition
    H, I = 512, 1376
    sizes = []
    for l in (1, 0):
        sizes += [(f"l{l}.down", H * I), (f"l{l}.gate", H * I), (f"l{l}.up", H * I), (f"l{l}.o", H * H), (f"l{l}.q", H * H), (f"l{l}.k", H * H), (f"l{l}.v", H * H), (f"l{l}.ln", H)]
    P = ZeroPartition(sizes, world, 0, H * H, no_close_after=[f"l{l}.{p}" for l in (0, 1) for p in "qk"])
    for l in (0, 1):
        assert P.offset[f"l{l}.k"] == P.offset[f"l{l}.q"] + H * H and P.offset[f"l{l}.v"] == P.offset[f"l{l}.q"] + 2 * H * H
    # the layout is still a partition: buckets tile [0, total), each splits evenly over the ranks, every tensor lies inside its bucket
    assert P.buckets[0][0] == 0 and P.buckets[-1][1] == P.total and all(P.buckets[i][1] == P.buckets[i + 1][0] for i in range(len(P.buckets) - 1))
    assert all((e - s) % world == 0 for s, e in P.buckets)
    for n, k in sizes:
        s, e = P.buckets[P.bucket_of[n]]
        assert s <= P.offset[n] and P.offset[n] + k <= e


def test_scheduler_bounds_the_rows_of_a_packed_prefill():
    """a packed prefill runs between two decode steps of the live requests, so its size bounds their stall.  The scheduler sizes a queued
    request as the reference's splice does (llava_arch.py:103-112: each image placeholder becomes num_patches rows) and stops filling a turn's job list
    at max_prefill_rows — but always takes one request."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "llava-plus-codebase_amd"))
    from llava_mi355x.batching import DecodeBatcher
    from llava_mi355x.constants import IMAGE_TOKEN_INDEX

    class _Tower: num_patches = 576
    class _Model:
        def get_vision_tower(self): return _Tower()
    class _Self: model = _Model()

    ids = torch.tensor([[1, 5, IMAGE_TOKEN_INDEX, 7, 9] + [3] * 507])
    rows = DecodeBatcher._request_rows(_Self(), {"ids": ids, "images": object()})
    assert rows == 512 - 1 + 576                                   # the headline request: 1087 positions
    assert DecodeBatcher._request_rows(_Self(), {"ids": ids[:, :4], "images": None}) == 4
    assert DecodeBatcher._request_rows(_Self(), {"ids": None}) == 1    # unsizable: the prefill itself will report it

    # the selection rule of DecodeBatcher._loop, restated on plain numbers
    def take(queue, cap_rows, cap_jobs=8):
        jobs, total = [], 0
        while queue and len(jobs) < cap_jobs:
            if jobs and cap_rows and total + queue[0] > cap_rows:
                break
            total += queue[0]; jobs.append(queue.pop(0))
        return jobs
    assert take([1087] * 5, 2048) == [1087]                        # 2 x 1087 > 2048: one per turn
    assert take([600, 600, 600, 600], 2048) == [600, 600, 600]
    assert take([5000, 10], 2048) == [5000]                        # an over-long request still goes, alone
    assert take([1087] * 9, 0) == [1087] * 8                       # 0 = unbounded: round 2's behaviour
