"""Continuous batching of decode steps.

The reference serves concurrent requests as independent `model.generate` threads with no batching
(llava/serve/model_worker.py:174-185, `--limit-model-concurrency` :264): every request streams all weights from HBM for
each of its tokens.  Here each request still prefills on its own thread and stream (chunked, so a long prompt does not
hold the GPU), then hands its sequence to ONE scheduler thread that advances every live sequence together through
`lmx_decode_batch`: one pass over the weights per step for all of them.  Requests join after their prefill and leave at
their stop condition, between any two steps.

    DecodeBatch     thin wrapper of lmx_batch_* (explicit batched steps: tests, bench, offline batch generation)
    DecodeBatcher   the scheduler (used by LlavaLlamaForCausalLM.generate once enable_batching() was called): the decode thread that steps the live
                    requests and — packed_prefill, single process — a prefill thread on a high-priority stream that prefills the requests waiting at that
                    moment TOGETHER (lmx_prefill_batch) beside the running decode steps and hands them over between two steps

Under tensor parallelism (one process per GPU) the scheduler of rank 0 is the LEADER: it is the only thread that issues work carrying
collectives — prefills included, which then run on the scheduler's stream between two decode steps instead of on the request's thread —
and it announces every such call on a command channel before making it, so the followers (tp_serving.py) replay the same calls in the
same order on their shards.
"""
from __future__ import annotations

import time
import ctypes
import threading
from typing import Callable, List, Optional, Sequence

import torch

from ._C import check, lib, ptr


class DecodeBatch:
    """Workspaces + device tables for up to `capacity` sequences stepping together."""

    def __init__(self, model, capacity: int = 32):
        self.model = model
        self.capacity = int(capacity)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(model.device):
            check(lib.lmx_batch_create(model._h, self.capacity, ctypes.byref(self._h)), "lmx_batch_create")

    def step(self, seqs: Sequence[ctypes.c_void_p], tokens: Optional[Sequence[int]] = None, n_steps: int = 1, greedy: bool = True,
             logits: Optional[torch.Tensor] = None, want_ids: bool = True) -> Optional[List[List[int]]]:
        """Advance every sequence by n_steps tokens.  Returns ids[step][member] (greedy picks) when want_ids."""
        n = len(seqs)
        arr = (ctypes.c_void_p * n)(*[s.value if isinstance(s, ctypes.c_void_p) else s for s in seqs])
        tk = None
        if tokens is not None:
            tk = (ctypes.c_int64 * n)(*[int(t) for t in tokens])
        ids = (ctypes.c_int64 * (n * n_steps))() if want_ids else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.model.device).cuda_stream)
        check(lib.lmx_decode_batch(self.model._h, self._h, arr, n, tk, int(n_steps), ptr(logits), int(bool(greedy)), ids, stream), "lmx_decode_batch")
        if not want_ids:
            return None
        return [[int(ids[s * n + i]) for i in range(n)] for s in range(n_steps)]

    def step_async(self, seqs: Sequence[ctypes.c_void_p], ids_pinned: torch.Tensor) -> None:
        """Enqueue one step; the picks land in `ids_pinned` (pinned int64 host tensor, >= len(seqs)) when the stream gets there."""
        n = len(seqs)
        arr = (ctypes.c_void_p * n)(*[s.value if isinstance(s, ctypes.c_void_p) else s for s in seqs])
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.model.device).cuda_stream)
        check(lib.lmx_decode_batch_async(self.model._h, self._h, arr, n, 1, ctypes.c_void_p(ids_pinned.data_ptr()), stream), "lmx_decode_batch_async")

    def close(self):
        if self._h:
            lib.lmx_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _Member:
    __slots__ = ("seq", "on_token", "done", "error", "room", "inflight", "finished", "rid", "cache", "request", "make_emit", "max_new")

    def __init__(self, seq, on_token, room):
        self.seq, self.on_token, self.room = seq, on_token, room
        self.done = threading.Event()
        self.error: Optional[BaseException] = None
        self.inflight = 0          # steps enqueued for this member whose picks were not processed yet
        self.finished = False
        self.rid = -1              # leader mode: request id on the command channel; the scheduler owns `cache`
        self.cache = None
        self.request = None
        self.make_emit = None
        self.max_new = 0


class DecodeBatcher:
    """One scheduler thread per model: steps all submitted sequences together until each one's `on_token` says stop.

    submit(seq, ...) is called by a request thread after its prefill (the sequence's first pick — argmax, or a draw when the
    sequence has sampling parameters, lmx_seq_set_sampling — is then on the device) and blocks until the request finished.
    `on_token(id) -> bool` runs on the scheduler thread (streamer put + stopping criteria of that request) and returns True
    to leave the batch.  Greedy and sampled requests mix freely: every member's pick happens inside the batched step."""

    def __init__(self, model, capacity: int = 32, channel=None, scheduler_prefill: bool = False, max_prefill_batch: int = 8, max_prefill_rows: int = 2304,
                 prefill_thread: int = 2):
        self.model = model
        self.capacity = int(capacity)
        self.channel = channel          # tensor-parallel leader: tp_serving.CommandChannel to the followers (None: single process)
        self._step_status = None        # tensor parallel: pending ok / fail exchange of the last announced decode step (CommandChannel.agree_begin)
        self.scheduler_prefill = bool(scheduler_prefill) or channel is not None      # requests are prefilled by the scheduler, several at a time
        # single process: the packed prefills run on a thread and a (high-priority) stream of their own, so a new request neither waits behind the decode
        # step in flight nor holds the live requests' next step back.  Under tensor parallelism every call that carries a collective must
        # be issued in ONE order on every rank: there the prefills stay between two decode steps on the leader's stream
        self.prefill_thread = bool(prefill_thread) and self.scheduler_prefill and channel is None
        # ... and the rank-local half of a request's prefill (image encode or feature-cache hit, splice, prefix-cache take: model._prepare_request) runs on the
        # request's OWN thread and stream before it queues, so the prefill thread only packs ready rows — and sizes a pack by the rows that are really left
        # after prefix reuse (prefill_thread=1 keeps that half on the prefill thread)
        self.prepare_on_request_thread = self.prefill_thread and int(prefill_thread) != 1
        self._prefilling = 0            # prefill-thread mode: requests taken off the queue whose prefill has not handed them to the decode loop yet
        self.max_prefill_batch = max(1, int(max_prefill_batch))
        # rows (prompt positions after the image splice) one packed prefill may hold: bounds how long the live requests wait between two of their
        # decode steps (a packed prefill runs between steps on the scheduler's stream; ~18 ms per 1k rows at 7B) and the packed workspace.  At least one
        # request is always taken, whatever its length; 0 = unbounded (round 2's behaviour: up to max_prefill_batch requests whatever their size)
        self.max_prefill_rows = max(0, int(max_prefill_rows))
        self.max_backlog_hold_s = 1.0   # longest the decode loop stands back for a burst's prefills (see _backlogged)
        self._hold_t0 = None
        self.prefill_batches = 0        # statistics: packed prefill calls / requests prefilled by them
        self.prefilled = 0
        self.batch = DecodeBatch(model, capacity)
        self._cv = threading.Condition()
        self._next_rid = 0
        self._requests: List[_Member] = []      # leader mode: requests whose prefill the scheduler thread still has to run
        self._waiting: List[_Member] = []
        self._live: List[_Member] = []          # members currently stepping (owned by the scheduler thread)
        self._stop = False
        self._paused = False
        self._broken: Optional[BaseException] = None     # tensor parallel: a step failed on some rank — the followers have left their loop, no collective may follow
        self.steps = 0                  # statistics: batched steps run / member-steps served
        self.member_steps = 0
        self.max_live = 0
        self._thread = threading.Thread(target=self._run, name="lmx-decode-batcher", daemon=True)
        self._thread.start()
        self._pf_thread = None
        if self.prefill_thread:
            self._pf_thread = threading.Thread(target=self._run_prefill, name="lmx-prefill-batcher", daemon=True)
            self._pf_thread.start()

    def submit(self, seq, on_token: Callable[[int], bool], room: int) -> None:
        m = _Member(seq, on_token, int(room))
        # the prefill ran on the caller's stream: it must be complete before the scheduler's stream touches the sequence
        torch.cuda.current_stream(self.model.device).synchronize()
        with self._cv:
            if self._broken is not None:
                raise RuntimeError(f"decode batcher is broken (tensor-parallel group lost): {self._broken}")
            if self._stop:
                raise RuntimeError("decode batcher is closed")
            self._waiting.append(m)
            self._cv.notify_all()
        m.done.wait()
        if m.error is not None:
            raise m.error

    def submit_request(self, request: dict, make_emit: Callable[[int], Callable[[int], bool]], max_new_tokens: int) -> None:
        """Hand a whole request (ids, images, mask, sampling, chunk) to the scheduler thread, which prefills it — together with the other requests waiting
        at that moment, announced to the followers under tensor parallelism — between two decode steps and then steps it with the others.
        make_emit(budget) builds the request's on_token.  Blocks until the request finished."""
        m = _Member(None, None, 0)
        m.request, m.make_emit, m.max_new = request, make_emit, int(max_new_tokens)
        torch.cuda.current_stream(self.model.device).synchronize()      # pixel values (and prepared rows) were put on the device by the caller's stream
        with self._cv:
            if self._broken is not None or self._stop:
                self._drop_prepared(m)
                if self._broken is not None:
                    raise RuntimeError(f"decode batcher is broken (tensor-parallel group lost): {self._broken}")
                raise RuntimeError("decode batcher is closed")
            m.rid = self._next_rid; self._next_rid += 1
            self._requests.append(m)
            self._cv.notify_all()
        m.done.wait()
        if m.error is not None:
            raise m.error

    def pause(self) -> None:
        """Stop taking steps (requests keep queueing); resume() continues.  Lets a test — or an operator draining a worker —
        line requests up so that they start decoding in the same step."""
        with self._cv:
            self._paused = True

    def resume(self) -> None:
        with self._cv:
            self._paused = False
            self._cv.notify_all()

    def queued(self) -> int:
        with self._cv:
            return len(self._waiting) + len(self._requests) + self._prefilling

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=30)
        if self._pf_thread is not None:
            self._pf_thread.join(timeout=30)
        if self.channel is not None and self._broken is None:
            self.channel.send(("stop",))
        self.batch.close()

    # ---- scheduler thread ------------------------------------------------------------------------------------------------
    def _run(self):
        try:
            self._loop()
        except BaseException as e:  # noqa: BLE001 — e.g. a device fault surfacing in an event wait: nobody may be left waiting
            with self._cv:
                self._stop = True
                stuck = list(self._live) + self._waiting + self._requests
                self._waiting = []; self._requests = []
            self._fail(stuck, e)

    def _loop(self):
        model = self.model
        live: List[_Member] = self._live
        try:
            torch.cuda.set_device(model.device)
            stream = torch.cuda.Stream(device=model.device)
        except BaseException as e:  # noqa: BLE001
            self._fail(live, e)
            return
        # Two-deep pipeline: step k+1 is enqueued BEFORE the host processes the picks of step k (streamer puts, stopping criteria,
        # tokenizer work of every member), so the GPU never waits for Python.  A member that stops at step k has then already
        # been stepped once more; that extra token is dropped (its KV slot was budgeted: `room` counts enqueued steps).
        pinned = [torch.empty((self.capacity,), dtype=torch.long).pin_memory() for _ in range(2)]
        events = [torch.cuda.Event() for _ in range(2)]
        pending = None                    # (members, slot)
        slot = 0
        with torch.cuda.stream(stream):
            while True:
                with self._cv:
                    while not self._stop and pending is None:
                        if self._paused or (not live and not self._waiting and (self.prefill_thread or not self._requests)):
                            self._cv.wait()
                        elif self._backlogged():
                            self._cv.wait(timeout=0.02)          # a burst is being prefilled: the prefill thread notifies after every pack
                        else:
                            break
                    hold = self._backlogged()                    # (with a step still in flight: launch no further one, its picks are processed below)
                    if self._stop:
                        stream.synchronize()
                        for m in live:
                            self._retire(m)
                        self._fail(live + self._waiting + self._requests, RuntimeError("decode batcher closed"))
                        self._waiting = []; self._requests = []
                        self._cv.notify_all()
                        return
                    while not self._paused and self._waiting and len(live) < self.capacity:        # join between steps
                        live.append(self._waiting.pop(0))
                    jobs = [] if self.prefill_thread else self._take_jobs(len(live))
                if jobs:
                    # one packed prefill per turn of the loop (the requests waiting right now), so live requests keep stepping between the prefills of a burst
                    self._leader_prefill(jobs, live)
                    if self._broken is not None:
                        stream.synchronize()
                        self._break(live, self._broken)
                        return
                self.max_live = max(self.max_live, len(live))
                launched = None
                go = [m for m in live if not m.finished and m.room - m.inflight > 0]
                if go and not self._paused and not hold:
                    try:
                        if self.channel is not None:
                            if not self.channel.agree_end(self._step_status):       # the previous step's ok / fail exchange (tp_serving.py)
                                self._step_status = None
                                raise RuntimeError("a decode step failed on another tensor-parallel rank")
                            self._step_status = None
                            self.channel.send(("step", [m.rid for m in go]))
                        ok = True
                        try:
                            self.batch.step_async([m.seq for m in go], pinned[slot])
                        except BaseException:  # noqa: BLE001
                            ok = False
                            raise
                        finally:
                            if self.channel is not None:
                                self._step_status = self.channel.agree_begin(ok)     # every rank reports after every announced step
                        events[slot].record(stream)
                        for m in go:
                            m.inflight += 1
                        launched = (go, slot)
                        slot ^= 1
                        self.steps += 1
                        self.member_steps += len(go)
                    except BaseException as e:  # noqa: BLE001
                        stream.synchronize()
                        if self.channel is not None:
                            # tensor parallel: every rank has seen (or will see at its next exchange) the failed step and the followers leave serve_follower.
                            # Any further send / agree would be a gloo collective against exited ranks: the batcher is dead from here on — fail everything
                            # queued, refuse new work, never touch the channel again
                            self._break(live, e)
                            return
                        self._fail(live, e)
                        live.clear(); pending = None
                        continue
                if pending is not None:
                    members, ps = pending
                    events[ps].synchronize()
                    ids = pinned[ps][: len(members)].tolist()
                    for i, m in enumerate(members):
                        m.inflight -= 1
                        if m.finished:
                            continue               # stopped at the previous step: the pipeline's extra step (the device-side stop rule left the sequence untouched)
                        if ids[i] < 0:
                            m.finished = True      # the device-side rule stopped it and the host's criteria did not: nothing more will come
                            continue
                        try:
                            m.room -= 1
                            if m.on_token(ids[i]) or m.room <= 0:
                                m.finished = True
                        except BaseException as e:  # noqa: BLE001
                            m.error = e
                            m.finished = True
                    done_now = [m for m in live if m.finished and m.inflight == 0]
                    live[:] = [m for m in live if not (m.finished and m.inflight == 0)]
                    for m in done_now:
                        self._retire(m)
                        m.done.set()
                    if done_now and self.prefill_thread:
                        with self._cv:
                            self._cv.notify_all()          # a slot is free: the prefill thread may be waiting for it
                pending = launched
                if pending is None:
                    # nothing in flight: members that finished with no step outstanding leave now
                    for m in [m for m in live if m.finished]:
                        self._retire(m)
                        m.done.set()
                    live[:] = [m for m in live if not m.finished]

    def _break(self, live: List[_Member], e: BaseException) -> None:
        """Tensor-parallel group lost: fail everything live or queued, refuse new work, never touch the channel again."""
        with self._cv:
            self._broken = e
            self._stop = True
            stuck = list(live) + self._waiting + self._requests
            self._waiting = []; self._requests = []
        for m in live:
            self._retire(m)
        self._fail(stuck, e)
        live.clear()

    def _request_rows(self, request: dict) -> int:
        """Prompt positions of a queued request after the image splice (llava_arch.py:103-112: every image placeholder becomes num_patches rows)."""
        try:
            if request.get("rows") is not None:
                return int(request["rows"])            # prepared on the request's thread: the rows left after prefix reuse
            ids = request["ids"]
            n = int(ids.numel())
            from .constants import IMAGE_TOKEN_INDEX
            n_img = int((ids == IMAGE_TOKEN_INDEX).sum().item()) if request.get("images") is not None else 0
            tower = self.model.get_vision_tower()
            return n + n_img * (int(getattr(tower, "num_patches", 576)) - 1)
        except Exception:  # noqa: BLE001 — an odd request is sized by the prefill itself
            return 1

    # ---- tensor-parallel leader ------------------------------------------------------------------------------------------
    def _leader_prefill(self, jobs: List[_Member], live: List[_Member]) -> None:
        """Announce the requests (tensor parallel), prefill them together on the scheduler's stream, deliver first tokens, let them join the live set."""
        caches = self._prefill_jobs(jobs)
        live.extend(self._first_tokens(jobs, caches))

    def _prefill_jobs(self, jobs: List[_Member]) -> list:
        """One packed prefill of `jobs` on the calling thread's stream.  Returns, per job, its LmxKVCache or the exception that failed it."""
        model = self.model
        try:
            chunk = max(m.request["prefill_chunk"] for m in jobs)
            if self.channel is not None:
                # tensor parallel: the pending step status first (same order of exchanges on every rank), then the announcement, then the prefill every
                # rank runs the same way — rank-local half, agreement, collective-bearing half, agreement (tp_serving.prefill_symmetric): a request
                # fails on all ranks or on none, and no rank re-runs collectives the others do not
                from .tp_serving import prefill_symmetric
                if not self.channel.agree_end(self._step_status):
                    self._step_status = None
                    self._broken = RuntimeError("a decode step failed on another tensor-parallel rank")
                    raise self._broken
                self._step_status = None
                self.channel.send(("prefill", [m.rid for m in jobs], [self.channel.wire_request(m.request) for m in jobs]))
                caches = prefill_symmetric(model, self.channel, [m.request for m in jobs], chunk)
            else:
                prepared = [m.request.pop("prepared", None) for m in jobs]         # from here this call owns the prepared sequences
                reqs = [dict(m.request, ids=m.request["ids"].to(model.device)) for m in jobs]
                try:
                    if all(p is not None for p in prepared):
                        # rows made ready on the request threads' streams (complete: submit_request synchronised them); they are consumed on this stream
                        try:
                            for p in prepared:
                                for t in (p["embeds"], p["valid"]):
                                    if isinstance(t, torch.Tensor) and t.is_cuda:
                                        t.record_stream(torch.cuda.current_stream(model.device))
                            model._run_prepared(prepared, chunk)
                            caches = [p["cache"] for p in prepared]
                        except BaseException:
                            for p in prepared:
                                p["cache"].close()
                            raise
                    else:
                        for p in prepared:
                            if p is not None:
                                p["cache"].close()
                        caches = model._prefill_requests(reqs, chunk)
                except BaseException:  # noqa: BLE001 — one bad request must not take its neighbours down: retry one by one (from the request, not from its prepared rows)
                    if len(jobs) == 1:
                        raise
                    caches = []
                    for m, r in zip(jobs, reqs):
                        try:
                            caches.append(model._prefill_requests([r], chunk)[0])
                        except BaseException as e:  # noqa: BLE001
                            caches.append(e)
            self.prefill_batches += 1; self.prefilled += len(jobs)
        except BaseException as e:  # noqa: BLE001
            caches = [e] * len(jobs)
            from .tp_serving import TensorParallelDesync
            if isinstance(e, TensorParallelDesync):
                self._broken = e                           # the group's collective order is lost: the scheduler loop fails everything live and stops (_break)
        return caches

    def _first_tokens(self, jobs: List[_Member], caches) -> List[_Member]:
        """Deliver the prefills' picks (token 1 of every request, read on the calling thread's stream) and return the members that go on decoding."""
        from ._C import stream_handle
        model = self.model
        go: List[_Member] = []
        for m, c in zip(jobs, caches):
            try:
                if isinstance(c, BaseException):
                    raise c
                m.cache = c
                m.seq = c.seqs[0]
                c.out_ref = m.request.get("out_ref")          # the request's emitted ids: what the prefix cache keys the sequence by when it is released
                m.request = None
                budget = min(m.max_new, model.s_max - lib.lmx_seq_length(m.seq))
                m.on_token = m.make_emit(budget)
                m.room = budget - 1
                first = (ctypes.c_int64 * 1)(); n1 = ctypes.c_int32(0)
                check(lib.lmx_seq_read_tokens(m.seq, first, 1, ctypes.byref(n1), stream_handle()), "read_tokens")
                if m.on_token(int(first[0])) or m.room <= 0:
                    m.finished = True
            except BaseException as e:  # noqa: BLE001
                m.error = e
                m.finished = True
            if m.finished:
                self._retire(m)
                m.done.set()
            else:
                go.append(m)
        return go

    # ---- prefill thread (single process) -----------------------------------------------------------------------------------
    def _take_jobs(self, admitted: int) -> List[_Member]:
        """Requests for the next packed prefill (called with _cv held): as many as are waiting, up to max_prefill_batch, the free slots and max_prefill_rows."""
        jobs, rows = [], 0
        while self._requests and not self._paused and admitted + len(jobs) < self.capacity and len(jobs) < self.max_prefill_batch:
            r = self._request_rows(self._requests[0].request)
            if jobs and self.max_prefill_rows and rows + r > self.max_prefill_rows:
                break                          # the rest of the burst goes with the next packed prefill
            rows += r
            jobs.append(self._requests.pop(0))
        return jobs

    def _backlogged(self) -> bool:
        """(prefill-thread mode, _cv held)  More than one full pack of prompt rows is queued BEHIND the pack in progress: a burst.  The decode loop then
        stands back until the backlog is worked off — the prefills run at the full speed of the GPU, as they did between two decode steps, and the
        median time to first token of a burst stays where it was (32 requests at once, 7B: 380 ms; 483 ms with the decode steps of the early requests sharing
        the GPU).  A trickle of short requests — the second turns of tool loops, a few dozen to a few hundred rows each — never gets there: decode steps and
        prefills keep running side by side."""
        if not self.prefill_thread or self._prefilling == 0 or not self._requests:
            self._hold_t0 = None
            return False
        rows = 0
        for m in self._requests:
            rows += self._request_rows(m.request)
            if rows > max(self.max_prefill_rows, 1):
                # the hold is BOUNDED: under sustained arrivals the backlog never drains, and live requests must not wait for it for seconds — after
                # `max_backlog_hold_s` of standing back the decode steps run beside the prefills again until the backlog has been worked off once
                now = time.monotonic()
                if self._hold_t0 is None:
                    self._hold_t0 = now
                return now - self._hold_t0 <= self.max_backlog_hold_s
        self._hold_t0 = None
        return False

    def _run_prefill(self):
        jobs: List[_Member] = []
        try:
            model = self.model
            torch.cuda.set_device(model.device)
            stream = torch.cuda.Stream(device=model.device, priority=-1)
            with torch.cuda.stream(stream):
                while True:
                    with self._cv:
                        while True:
                            if self._stop:
                                return
                            jobs = self._take_jobs(len(self._live) + len(self._waiting) + self._prefilling)
                            if jobs:
                                self._prefilling += len(jobs)
                                break
                            self._cv.wait()
                    try:
                        caches = self._prefill_jobs(jobs)
                        go = self._first_tokens(jobs, caches)        # reads the picks: the prefill stream is idle behind it
                    except BaseException as e:  # noqa: BLE001
                        go = []
                        self._fail([m for m in jobs if not m.done.is_set()], e)
                    with self._cv:
                        self._prefilling -= len(jobs)
                        jobs = []
                        if self._stop:
                            for m in go:
                                self._retire(m)
                            self._fail(go, RuntimeError("decode batcher closed"))
                        else:
                            self._waiting.extend(go)                   # they join the running batch between two of its steps
                        self._cv.notify_all()
        except BaseException as e:  # noqa: BLE001 — nobody may be left waiting
            with self._cv:
                self._stop = True
                stuck = jobs + self._requests
                self._requests = []
                self._cv.notify_all()
            self._fail([m for m in stuck if not m.done.is_set()], e)

    def _retire(self, m: _Member) -> None:
        """Leader mode: the scheduler owns the member's sequence; tell the followers to drop theirs."""
        if m.rid < 0:
            return                                  # a member whose request thread owns its sequence (submit())
        try:
            if self.channel is not None and self._broken is None:
                self.channel.send(("release", m.rid))
        finally:
            if m.cache is not None:
                self.model._release_request_cache(m.cache)     # prefix cache (enable_reuse) or the engine's pool
                m.cache = None

    @staticmethod
    def _drop_prepared(m) -> None:
        """A queued request that never reaches a prefill: the sequence its own thread prepared for it (model._prepare_request) goes back to the pool."""
        p = m.request.pop("prepared", None) if isinstance(m.request, dict) else None
        if p is not None:
            try:
                p["cache"].close()
            except Exception:  # noqa: BLE001
                pass

    @staticmethod
    def _fail(members, e):
        for m in members:
            DecodeBatcher._drop_prepared(m)
            m.error = e
            m.done.set()
