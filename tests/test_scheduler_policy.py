"""CPU: the packing and backlog rules of the serving scheduler (llava_mi355x/batching.py: DecodeBatcher._take_jobs / _backlogged) on a stand-in object — no GPU, no
threads: which queued requests form the next packed prefill, and when the decode loop stands back for a burst."""
import os
import sys
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-plus-codebase_amd"))


def _stub(rows, capacity=32, max_prefill_batch=8, max_prefill_rows=2304, prefilling=0, paused=False):
    from llava_mi355x.batching import DecodeBatcher
    s = types.SimpleNamespace(capacity=capacity, max_prefill_batch=max_prefill_batch, max_prefill_rows=max_prefill_rows, prefill_thread=True, _paused=paused,
                              _prefilling=prefilling, _requests=[types.SimpleNamespace(request={"rows": r}) for r in rows], max_backlog_hold_s=1.0, _hold_t0=None)
    s._request_rows = lambda req: DecodeBatcher._request_rows(s, req)
    s.take = lambda admitted=0: DecodeBatcher._take_jobs(s, admitted)
    s.backlogged = lambda: DecodeBatcher._backlogged(s)
    return s


def test_a_pack_holds_what_fits_the_row_bound_the_batch_bound_and_the_free_slots():
    s = _stub([1087, 1087, 1087, 1087])
    assert [m.request["rows"] for m in s.take()] == [1087, 1087] and len(s._requests) == 2        # 3 x 1087 > 2304 rows
    s = _stub([5000, 100])
    assert [m.request["rows"] for m in s.take()] == [5000]                                         # one request is always taken, whatever its length
    s = _stub([120] * 12)
    assert len(s.take()) == 8 and len(s._requests) == 4                                            # max_prefill_batch
    s = _stub([120] * 12)
    assert len(s.take(admitted=29)) == 3                                                           # 32 slots, 29 taken
    assert s.take(admitted=32) == []
    assert _stub([120] * 4, paused=True).take() == []
    # second turns of tool loops are sized by the rows LEFT after prefix reuse (prepared on their own threads): 19 of them fit one row bound, 8 the batch bound
    s = _stub([120] * 19, max_prefill_batch=32)
    assert len(s.take()) == 19


def test_the_decode_loop_stands_back_only_for_a_burst():
    assert _stub([1087] * 30, prefilling=2).backlogged()                  # 30 single-image prompts behind the pack in progress
    assert not _stub([1087] * 30, prefilling=0).backlogged()              # nothing in progress (e.g. no free slot): decode goes on
    assert not _stub([120] * 6, prefilling=3).backlogged()                # a trickle of short second turns: decode steps and prefills side by side
    assert not _stub([], prefilling=2).backlogged()
    assert _stub([1200, 1200], prefilling=1).backlogged() and not _stub([1200, 1000], prefilling=1).backlogged()       # strictly more than one pack of rows


def test_the_hold_is_bounded_under_sustained_arrivals():
    """a backlog that never drains must not keep live requests from stepping for seconds — the hold lasts at most max_backlog_hold_s, then decode steps
    run beside the prefills until the backlog has been worked off once (which re-arms the hold)."""
    import time
    s = _stub([1087] * 30, prefilling=2)
    s.max_backlog_hold_s = 0.05
    assert s.backlogged() and s.backlogged()
    time.sleep(0.08)
    assert not s.backlogged() and not s.backlogged()                        # still backlogged, but the hold has run out
    s._requests = []                                                        # worked off
    assert not s.backlogged() and s._hold_t0 is None
    s._requests = [types.SimpleNamespace(request={"rows": 1087}) for _ in range(30)]
    assert s.backlogged()                                                   # a new burst: a new hold
