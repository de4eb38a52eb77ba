"""Tensor-parallel serving: one process per GPU, rank 0 serves, the others follow (BASELINE config 4).

The reference spreads a model over GPUs inside ONE process (`device_map="auto"`, llava/model/builder.py:26-34) and its worker threads
call `model.generate` concurrently (llava/serve/model_worker.py:174-185).  With one process per GPU and tensor-parallel shards every
rank must make the SAME engine calls in the SAME order — each decoder layer carries all-reduces — while only rank 0 sees the HTTP
requests, runs the tokenizer and decides when a request stops.  So rank 0 (the LEADER) funnels every collective-bearing call through
its scheduler thread (batching.DecodeBatcher with a channel) and announces each one on a command channel before making it:

    ("prefill", [rid...], [request...])   image encode + splice + PACKED prefill of the requests waiting at that moment (ids, pixel values, mask, sampling + SEED, chunk)
    ("step",    [rid, ...])     one batched decode step over these live requests, in this member order
    ("release", rid)            the request left the batch: free its sequence
    ("stop",)                   the leader closed its scheduler

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

    @staticmethod
    def wire_request(req: dict) -> dict:
        """Host copy of a request for the wire (pixel values may live on the leader's GPU)."""
        def host(x):
            if isinstance(x, torch.Tensor):
                return x.detach().cpu()
            if isinstance(x, (list, tuple)):
                return [host(v) for v in x]
            return x
        return {k: host(v) for k, v in req.items()}


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
                if kind == "stop":
                    break
                if kind == "prefill":
                    _, rids, reqs = cmd
                    reqs = [dict(r, ids=r["ids"].to(model.device), images=to_dev(r["images"])) for r in reqs]
                    chunk = max(r["prefill_chunk"] for r in reqs)
                    try:
                        for rid, c in zip(rids, model._prefill_requests(reqs, chunk)):
                            caches[rid] = c
                    except BaseException:  # noqa: BLE001 — the leader fails the same way, retries one by one and releases what stays broken
                        for rid, r in zip(rids, reqs):
                            try:
                                caches[rid] = model._prefill_requests([r], chunk)[0]
                            except BaseException:  # noqa: BLE001
                                stats["errors"] += 1
                    stats["prefill"] += len(rids)
                elif kind == "step":
                    seqs = [caches[r].seqs[0] for r in cmd[1]]
                    if events[slot] is not None:
                        events[slot].synchronize()          # the picks buffer of two steps ago is free again
                    batch.step_async(seqs, pinned[slot])
                    ev = torch.cuda.Event(); ev.record(stream); events[slot] = ev
                    slot ^= 1
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
