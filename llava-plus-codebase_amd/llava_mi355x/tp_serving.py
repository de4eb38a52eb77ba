"""Tensor-parallel serving: one process per GPU, rank 0 serves, the others follow (BASELINE config 4).

The reference spreads a model over GPUs inside ONE process (`device_map="auto"`, llava/model/builder.py:26-34) and its worker threads
call `model.generate` concurrently (llava/serve/model_worker.py:174-185).  With one process per GPU and tensor-parallel shards every
rank must make the SAME engine calls in the SAME order — each decoder layer carries all-reduces — while only rank 0 sees the HTTP
requests, runs the tokenizer and decides when a request stops.  So rank 0 (the LEADER) funnels every collective-bearing call through
its scheduler thread (batching.DecodeBatcher with a channel) and announces each one on a command channel before making it:

    ("prefill", [rid...], [request...])   image encode + splice + PACKED prefill of the requests waiting at that moment (ids, pixel values, mask, sampling + SEED, chunk,
                                          the id rules of the stop test: every rank's pick kernel stops the sequence at the same token, lmx_seq_set_stop)
    ("step",    [rid, ...])     one batched decode step over these live requests, in this member order
    ("release", rid)            the request left the batch: free its sequence
    ("stop",)                   the leader closed its scheduler

Failure handling (every rank must keep making the same collective-bearing calls): after a "prefill" command all ranks run the rank-local half of every
request, AGREE on which ones every rank could prepare (`CommandChannel.agree`, a tiny all-reduce on the CPU group), run the collective-bearing half
over the agreed ones and agree again on its outcome (`prefill_symmetric`); after every "step" they exchange an ok / fail flag asynchronously
(`agree_begin` / `agree_end`, finished before the next command).  A request therefore fails on all ranks or on none, and a rank that cannot launch a
step is noticed by the others at the next command instead of leaving them inside an all-reduce for ever.

Followers (`serve_follower`) replay the calls on their shards.  They never look at the picks: logits are identical on all ranks after
the all-reduce / vocabulary gather, a sampled request draws from the seed the leader put in the command, and stop decisions arrive as
"release".  The channel is a CPU (gloo) process group: commands are tiny except for the pixel values of a new request.

    rank 0:      model.enable_batching(capacity, channel=CommandChannel(group))   then serve as usual (model_worker.py unchanged)
    rank 1..N-1: serve_follower(model, CommandChannel(group), capacity)            returns at "stop"
"""
from __future__ import annotations

import ctypes
from typing import Dict

import torch

from ._C import check, lib, stream_handle
from .batching import DecodeBatch


class CommandChannel:
    """Ordered broadcast of small python objects from rank `src` to every rank of a torch.distributed group (use a gloo group: the
    payload is host data, and the group must not be shared with other traffic while a scheduler is running)."""

    def __init__(self, group=None, src: int = 0):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("tensor-parallel serving needs torch.distributed (the command channel is a broadcast from rank 0)")
        self.dist, self.group, self.src = dist, group, int(src)
        self.sent = 0

    def send(self, cmd) -> None:
        self.dist.broadcast_object_list([cmd], src=self.src, group=self.group)
        self.sent += 1

    def recv(self):
        box = [None]
        self.dist.broadcast_object_list(box, src=self.src, group=self.group)
        return box[0]

    def agree(self, flags) -> list:
        """Element-wise AND of per-rank ok flags (one small all-reduce on the CPU group).  EVERY rank calls it at the same point of the command
        stream; afterwards all ranks hold the same verdicts, so a rank-local failure is handled identically everywhere instead of leaving the
        others inside a collective the failed rank never enters."""
        t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return [bool(v) for v in t.tolist()]

    def agree_begin(self, ok: bool):
        """Asynchronous form for the decode steps: started right after a step is launched, finished (agree_end) before the next command, so the
        exchange overlaps the GPU work of the step."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        return t, self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group, async_op=True)

    @staticmethod
    def agree_end(pending) -> bool:
        if pending is None:
            return True
        t, work = pending
        work.wait()
        return bool(int(t[0]))

    @staticmethod
    def wire_request(req: dict) -> dict:
        """Host copy of a request for the wire (pixel values may live on the leader's GPU)."""
        def host(x):
            if isinstance(x, torch.Tensor):
                return x.detach().cpu()
            if isinstance(x, (list, tuple)):
                return [host(v) for v in x]
            return x
        return {k: host(v) for k, v in req.items() if k != "out_ref"}       # out_ref: the leader's own list of emitted ids (prefix-cache key), not for the wire


class TensorParallelDesync(RuntimeError):
    """The ranks of a tensor-parallel group no longer agree on their collective order: stop serving on this group (DecodeBatcher._break; followers leave)."""


def prefill_symmetric(model, channel: CommandChannel, reqs, chunk: int) -> list:
    """The announced prefill as EVERY rank runs it (leader's scheduler thread and followers alike).  Returns one entry per request: its LmxKVCache, or
    the exception that failed it — the SAME requests fail on every rank:
      0. (two or more requests) the requests' images go through one tower pass split over the ranks — see the comment in the body;
      1. every rank runs the rank-local half of each request (image encode, splice, sequence allocation: model._prepare_request) — the steps that can
         fail on one rank alone (a device copy, an allocation);
      2. the ranks agree (CommandChannel.agree) which requests every rank could prepare; the others are dropped everywhere;
      3. the collective-bearing half (decoder prefill with its all-reduces: model._run_prepared) runs over the agreed requests, packed, identically on
         every rank, and the ranks agree once more on its outcome (an error there is raised by argument checks BEFORE any launch, i.e. on every rank
         or on none; the second exchange turns the remaining case into a symmetric failure instead of a hang at the next collective)."""
    # 0. several requests with images: ONE tower pass over all of them, the images split over the ranks and the features all-gathered
    #    (model._preencode_requests).  It carries a collective, so it is bracketed like step 3: first agree which requests' pixel
    #    tensors every rank holds with the tower's geometry, run the pass over exactly those, agree on its outcome.
    feats = [None] * len(reqs)
    if getattr(model, "tp_world", 1) > 1 and len(reqs) > 1:
        def usable(r):
            try:
                im = r["images"]
                if not (isinstance(im, torch.Tensor) and im.dim() == 4 and im.shape[0] > 0):
                    return False
                model._check_pixels(im)
                return True
            except BaseException:  # noqa: BLE001
                return False
        use = channel.agree([usable(r) for r in reqs])
        # rank-local half first (checks, concatenation, allocations, this rank's share of the tower: model._preencode_local) — a failure here is agreed on
        # BEFORE any rank has queued a collective, so it fails the requests symmetrically and the group keeps serving
        st, loc_err = None, None
        try:
            st = model._preencode_local([r["images"] if u else None for r, u in zip(reqs, use)])
        except BaseException as e:  # noqa: BLE001
            loc_err = e
        if not channel.agree([loc_err is None])[0]:
            err = loc_err if loc_err is not None else RuntimeError("the image encode could not be prepared on another tensor-parallel rank")
            return [err for _ in reqs]
        if st is not None:
            # collective half: only all-reduces on buffers that exist.  A disagreement AFTER it means some rank's stream holds a collective the others do not
            # match: the group's collective order is lost, so this is not a per-request error — the caller stops serving (TensorParallelDesync)
            enc_err = None
            try:
                feats = model._preencode_collective(st)
            except BaseException as e:  # noqa: BLE001
                enc_err = e
            if not channel.agree([enc_err is None])[0]:
                model.clear_image_cache()
                raise TensorParallelDesync("the sharded image encode failed on a tensor-parallel rank after its collective was queued: "
                                           + repr(enc_err if enc_err is not None else "another rank"))
    prepared = []
    for r, f in zip(reqs, feats):
        try:
            imgs = r["images"]
            kw = {} if f is None else {"feats": f}
            prepared.append(model._prepare_request(r["ids"].to(model.device), imgs, r["attention_mask"], r["sampling"], r.get("stop"), **kw))
        except BaseException as e:  # noqa: BLE001
            prepared.append(e)
    ok_all = channel.agree([not isinstance(p, BaseException) for p in prepared])
    out = []
    for p, ok in zip(prepared, ok_all):
        if ok:
            out.append(p)
        else:
            if not isinstance(p, BaseException):
                p["cache"].close()
                p = RuntimeError("another tensor-parallel rank could not prepare this request")
            out.append(p)
    good = [p for p in out if not isinstance(p, BaseException)]
    run_err = None
    if good:
        try:
            model._run_prepared(good, chunk)
        except BaseException as e:  # noqa: BLE001
            run_err = e
    if not channel.agree([run_err is None])[0]:
        err = run_err if run_err is not None else RuntimeError("the packed prefill failed on another tensor-parallel rank")
        for p in good:
            p["cache"].close()
        return [p if isinstance(p, BaseException) else err for p in out]
    return [p if isinstance(p, BaseException) else p["cache"] for p in out]


def serve_follower(model, channel: CommandChannel, capacity: int = 32, on_command=None, record_tokens: bool = False) -> Dict[str, int]:
    """Replay the leader's calls until it says stop.  Returns counters (commands by kind) for logs and tests; record_tokens=True also
    reads each sequence's picks back when it is released (stats["tokens"][rid]; a test's proof that this rank drew what the leader
    streamed — costs a stream sync per request)."""
    model._ensure_final()
    batch = DecodeBatch(model, capacity)
    caches: Dict[int, object] = {}
    stats = {"prefill": 0, "step": 0, "release": 0, "errors": 0}
    stream = torch.cuda.Stream(device=model.device)
    pinned = [torch.empty((int(capacity),), dtype=torch.long).pin_memory() for _ in range(2)]
    events: list = [None, None]
    slot = 0
    step_status = None                  # pending ok / fail exchange of the last decode step (CommandChannel.agree_begin)

    def to_dev(x):
        if isinstance(x, torch.Tensor):
            return x.to(model.device)
        if isinstance(x, list):
            return [to_dev(v) for v in x]
        return x

    try:
        with torch.cuda.device(model.device), torch.cuda.stream(stream):
            while True:
                cmd = channel.recv()
                if on_command is not None:
                    on_command(cmd)
                kind = cmd[0]
                if not channel.agree_end(step_status):
                    raise RuntimeError("a decode step failed on another tensor-parallel rank: this rank's stream may be waiting in a collective; leaving")
                step_status = None
                if kind == "stop":
                    break
                if kind == "prefill":
                    _, rids, reqs = cmd

                    def local(r):
                        try:
                            return dict(r, images=to_dev(r["images"]))
                        except BaseException:  # noqa: BLE001 — reported through the agreement inside prefill_symmetric
                            return dict(r, images=None, ids=None)
                    chunk = max(r["prefill_chunk"] for r in reqs)
                    for rid, c in zip(rids, prefill_symmetric(model, channel, [local(r) for r in reqs], chunk)):
                        if isinstance(c, BaseException):
                            stats["errors"] += 1            # failed on every rank alike: the leader reports it to the client and sends "release"
                        else:
                            caches[rid] = c
                    stats["prefill"] += len(rids)
                elif kind == "step":
                    missing = [r for r in cmd[1] if r not in caches]
                    if missing:
                        # cannot happen after the agreed prefill; if it does, the ranks' views have diverged and replaying would hang in a collective
                        channel.agree_begin(False)
                        raise RuntimeError(f"tensor-parallel follower: step names unknown request(s) {missing}; the leader is told through the step status")
                    seqs = [caches[r].seqs[0] for r in cmd[1]]
                    ok = True
                    try:
                        if events[slot] is not None:
                            events[slot].synchronize()          # the picks buffer of two steps ago is free again
                        batch.step_async(seqs, pinned[slot])
                        ev = torch.cuda.Event(); ev.record(stream); events[slot] = ev
                        slot ^= 1
                    except BaseException:  # noqa: BLE001
                        ok = False
                    step_status = channel.agree_begin(ok)
                    if not ok:
                        raise RuntimeError("tensor-parallel follower: a decode step could not be launched on this rank")
                    stats["step"] += 1
                elif kind == "release":
                    c = caches.pop(cmd[1], None)
                    if c is not None:
                        if record_tokens:
                            host = (ctypes.c_int64 * model.s_max)(); n = ctypes.c_int32(0)
                            check(lib.lmx_seq_read_tokens(c.seqs[0], host, model.s_max, ctypes.byref(n), stream_handle()), "read_tokens")
                            stats.setdefault("tokens", {})[int(cmd[1])] = [int(host[i]) for i in range(n.value)]
                        c.close()
                    stats["release"] += 1
                else:
                    raise RuntimeError(f"unknown command {kind!r} on the tensor-parallel channel")
            stream.synchronize()
    finally:
        for c in caches.values():
            c.close()
        batch.close()
    return stats
