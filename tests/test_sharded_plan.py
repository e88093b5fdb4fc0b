"""The pipeline of svsx_sharded_search_device, checked for ORDER without a GPU.

`svsx_sharded_plan` (include/svsx.h) lists every operation one call enqueues on its three streams -- the library executes exactly that list
(csrc/svsx_sharded.hip) -- and this test replays it through a happens-before model of HIP streams and events:
  * operations of one stream run in enqueue order;
  * hipStreamWaitEvent(stream, e) orders everything enqueued on `stream` afterwards behind the LAST record of `e` enqueued before the wait
    (an event never recorded in this call stands for the previous call's record);
and asserts what the data path needs (every read after its write), what the pipelining claims (the second chunk's searches are NOT ordered
behind the first chunk's exchange), and that the call is "enqueued on the caller's stream" (its last operation there is behind everything).
"""
import ctypes as C

import pytest

from scalablevectorsearch_amd import _lib

WAIT, RECORD, SEARCH, LOCAL_MERGE, ALLGATHER, WORLD_MERGE, STAMP = range(7)
EV_START, EV_JOIN0, EV_JOIN1, EV_READY0, EV_READY1, EV_DONE0, EV_DONE1, EV_LAST = range(8)


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("stream", C.c_int32), ("chunk", C.c_int32), ("arg", C.c_int32)]


def plan(n_local, exchange, nq, have_last):
    L = _lib.lib()
    L.svsx_sharded_plan.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.POINTER(Op), C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t(0)
    assert L.svsx_sharded_plan(n_local, int(exchange), nq, int(have_last), None, 0, C.byref(n)) == 0
    ops = (Op * n.value)()
    assert L.svsx_sharded_plan(n_local, int(exchange), nq, int(have_last), ops, n.value, C.byref(n)) == 0
    return [(o.kind, o.stream, o.chunk, o.arg) for o in ops]


def happens_before(ops):
    """before[i] = set of operation indices that are guaranteed complete when operation i starts"""
    last_on_stream = {}
    last_record = {}       # event -> index of its latest RECORD
    pending = {0: set(), 1: set(), 2: set()}  # per stream: operations every later operation of that stream is ordered behind (through waits)
    before = []
    for i, (kind, stream, chunk, arg) in enumerate(ops):
        b = set(pending[stream])
        if stream in last_on_stream:
            j = last_on_stream[stream]
            b |= before[j] | {j}
        if kind == WAIT and arg in last_record:
            j = last_record[arg]
            b |= before[j] | {j}
        before.append(b)
        pending[stream] = set(b)
        last_on_stream[stream] = i
        if kind == RECORD:
            last_record[arg] = i
    return before


CASES = [(n_local, exchange, nq, have_last) for n_local in (1, 2, 3, 8) for exchange in (0, 1) for nq in (100, 8191, 8192, 10_000) for have_last in (0, 1)]


@pytest.mark.parametrize("n_local,exchange,nq,have_last", CASES)
def test_every_read_follows_its_write(n_local, exchange, nq, have_last):
    ops = plan(n_local, exchange, nq, have_last)
    hb = happens_before(ops)
    nchunk = 2 if (exchange and nq >= 8192) else 1
    idx = lambda kind, chunk: [i for i, o in enumerate(ops) if o[0] == kind and o[2] == chunk]
    assert sorted({o[2] for o in ops if o[0] == SEARCH}) == list(range(nchunk))
    for c in range(nchunk):
        searches = idx(SEARCH, c)
        assert sorted(ops[i][3] for i in searches) == list(range(n_local))        # every local shard, once per chunk
        lm, ag, wm = idx(LOCAL_MERGE, c), idx(ALLGATHER, c), idx(WORLD_MERGE, c)
        assert len(lm) == (1 if n_local > 1 else 0) and len(ag) == len(wm) == (1 if exchange else 0)
        if lm:
            assert set(searches) <= hb[lm[0]]                                       # the local merge reads every shard's list
        if ag:
            assert set(searches + lm) <= hb[ag[0]]                                  # the send slot is complete before it is gathered
            assert ag[0] in hb[wm[0]]                                               # the world merge reads the gathered buffer
    # the caller's stream: its last operation is the record of ev_last, and it is ordered behind EVERYTHING the call enqueued anywhere
    last = len(ops) - 1
    assert ops[last][:2] == (RECORD, 0) and ops[last][3] == EV_LAST
    assert hb[last] == set(range(last))
    # chained behind the previous call's staging when there was one: the very first operation, on the caller's stream
    assert (ops[0] == (WAIT, 0, -1, EV_LAST)) == bool(have_last)
    if have_last:  # ... and nothing of this call can start before it: every operation is ordered behind that wait
        assert all(0 in hb[i] for i in range(1, len(ops)))
    # the send slot / gather buffer of chunk c are distinct regions per chunk, but the SAME chunk's staging is never written twice
    for c in range(nchunk):
        for i in idx(SEARCH, c):
            assert not any(j in hb[i] for j in idx(WORLD_MERGE, c))                  # (sanity: no search of a chunk behind its own merge)


@pytest.mark.parametrize("n_local", [1, 2, 8])
def test_the_second_chunk_searches_under_the_first_chunks_exchange(n_local):
    """batches of >= 8 192 queries with an exchange: the all-gather + world merge of chunk 0 must not hold up the searches of chunk 1"""
    ops = plan(n_local, 1, 10_000, 0)
    hb = happens_before(ops)
    ag0 = next(i for i, o in enumerate(ops) if o[0] == ALLGATHER and o[2] == 0)
    wm0 = next(i for i, o in enumerate(ops) if o[0] == WORLD_MERGE and o[2] == 0)
    s1 = [i for i, o in enumerate(ops) if o[0] == SEARCH and o[2] == 1]
    assert s1 and all(ag0 not in hb[i] and wm0 not in hb[i] for i in s1)
    # one local shard: the exchange lives on an internal stream; the caller's stream only joins it at the very end
    if n_local == 1:
        assert ops[ag0][1] == 1 and all(ops[i][1] == 0 for i in s1)
        joins = [i for i, o in enumerate(ops) if o[0] == WAIT and o[1] == 0 and o[3] in (EV_DONE0, EV_DONE1)]
        assert len(joins) == 2 and min(joins) > max(s1)
    else:  # several local shards: their searches alternate between the two internal streams, both in use
        assert {ops[i][1] for i in s1} == {1, 2}


def test_small_batches_and_single_rank_objects_stay_simple():
    ops = plan(1, 0, 10_000, 0)  # one shard, no exchange: one search on the caller's stream, nothing else but the stamps and ev_last
    assert [o for o in ops if o[0] not in (STAMP,)] == [(SEARCH, 0, 0, 0), (RECORD, 0, -1, EV_LAST)]
    ops = plan(1, 1, 100, 0)     # exchange, one chunk: everything on the caller's stream, in data order
    assert [o[0] for o in ops if o[0] != STAMP] == [SEARCH, ALLGATHER, WORLD_MERGE, RECORD] and {o[1] for o in ops} == {0}
    # stamps: five phase boundaries per chunk, in order, each exactly once
    for args in ((1, 1, 10_000, 1), (4, 1, 10_000, 0), (4, 0, 500, 0)):
        ops = plan(*args)
        for c in sorted({o[2] for o in ops if o[0] == STAMP}):
            assert [o[3] for o in ops if o[0] == STAMP and o[2] == c] == ([0, 1, 2, 3, 4] if args[1] else [0, 1, 2])


def test_plan_argument_errors():
    L = _lib.lib()
    n = C.c_size_t(0)
    assert L.svsx_sharded_plan(0, 0, 10, 0, None, 0, C.byref(n)) == _lib.INVALID_ARGUMENT
    assert L.svsx_sharded_plan(1, 0, 0, 0, None, 0, C.byref(n)) == _lib.INVALID_ARGUMENT
    ops = (Op * 1)()
    assert L.svsx_sharded_plan(2, 1, 10_000, 0, ops, 1, C.byref(n)) == _lib.INVALID_ARGUMENT and n.value > 1
