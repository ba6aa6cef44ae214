"""feeder.ReaderPool (forked reader processes that pack batches into shared slots) without a GPU: order, contents, slots that grow,
errors and dead workers.  The device half (registered slots, one DMA per batch) is tests/test_h5_boundary_gpu.py's."""
import os
import signal

import numpy as np
import pytest

from himo_amd.feeder import ReaderPool, _Ref, _swap_refs


class _Batch:
    def __init__(self, k, pts, ids, empty):
        self.k, self.pts, self.ids, self.empty = k, pts, ids, empty


def _make(k):
    rng = np.random.default_rng(k)
    return [rng.normal(size=(int(rng.integers(50, 400)), 3)).astype(np.float32) for _ in range(3)], k


def _build(item, upload):
    parts, k = item
    ids = [np.full(len(p), i, dtype=np.uint32) for i, p in enumerate(parts)]
    return _Batch(k, upload(parts, np.float32), upload(ids, np.int64), upload([np.empty((0, 3), np.float32)], np.float32))


def _host_view(slot, used):
    buf = slot.numpy()[:used]

    def view(ref):
        n = int(np.prod(ref.shape)) * np.dtype(ref.dtype).itemsize
        return buf[ref.off:ref.off + n].view(np.dtype(ref.dtype)).reshape(ref.shape).copy()
    return view


def _check(k, obj):
    parts, _ = _make(k)
    assert obj.k == k
    np.testing.assert_array_equal(obj.pts, np.concatenate(parts))
    assert obj.ids.dtype == np.int64 and obj.ids.tolist() == [i for i, p in enumerate(parts) for _ in range(len(p))]
    assert obj.empty.shape == (0, 3)


@pytest.mark.parametrize("workers", [1, 3])
def test_batches_come_back_in_order_with_their_arrays(workers):
    pool = ReaderPool(11, _make, _build, workers=workers, slot_bytes=1 << 20)
    seen = []
    for k, s, used, obj in pool:
        assert isinstance(obj.pts, _Ref) and obj.pts.off % 64 == 0 and obj.ids.off % 64 == 0
        _swap_refs(obj, _host_view(pool.slots[s], used))
        _check(k, obj)
        seen.append(k)
        pool.release(s)
    assert seen == list(range(11)) and pool.restarts == 0
    assert pool.slots == [] and pool._procs == []                      # workers joined, mappings closed


def test_a_batch_that_does_not_fit_restarts_with_larger_slots():
    registered, unregistered = [], []
    pool = ReaderPool(9, _make, _build, workers=2, slot_bytes=4096,     # a batch needs ~ 3 x 225 x (12 + 8) bytes
                      on_slots=lambda ts: registered.append([t.numel() for t in ts]), off_slots=lambda ts: unregistered.append(len(ts)))
    seen = []
    for k, s, used, obj in pool:
        _swap_refs(obj, _host_view(pool.slots[s], used))
        _check(k, obj)
        seen.append(k)
        pool.release(s)
    assert seen == list(range(9))
    assert pool.restarts >= 1 and pool.slot_bytes > 4096
    assert len(registered) == pool.restarts + 1 == len(unregistered)    # every generation of slots registered once, unregistered once
    assert registered[0][0] == 4096 and registered[-1][0] == pool.slot_bytes


def test_without_restart_a_batch_that_does_not_fit_is_an_error_naming_the_size():
    pool = ReaderPool(5, _make, _build, workers=2, slot_bytes=4096, restart=False)
    with pytest.raises(RuntimeError, match=r"slot_bytes >= \d+"):
        for k, s, used, obj in pool:
            pool.release(s)
    assert pool.restarts == 0 and pool._procs == []


def test_a_readers_exception_reaches_the_consumer():
    def make(k):
        if k == 4:
            raise KeyError("seflowpp_best")
        return _make(k)
    pool = ReaderPool(8, make, _build, workers=2, slot_bytes=1 << 20)
    got = []
    with pytest.raises(KeyError, match="seflowpp_best"):
        for k, s, used, obj in pool:
            got.append(k)
            pool.release(s)
    assert got == [0, 1, 2, 3]
    assert pool._procs == []


def test_a_dead_reader_is_an_error_not_a_hang():
    def make(k):
        if k == 2:
            os.kill(os.getpid(), signal.SIGKILL)
        return _make(k)
    pool = ReaderPool(6, make, _build, workers=2, slot_bytes=1 << 20)
    with pytest.raises(RuntimeError, match="died"):
        for k, s, used, obj in pool:
            pool.release(s)


def test_stopping_early_and_an_empty_list():
    assert list(ReaderPool(0, _make, _build)) == []
    pool = ReaderPool(50, _make, _build, workers=2, slot_bytes=1 << 20)
    for k, s, used, obj in pool:
        pool.release(s)
        if k == 3:
            break
    assert pool._procs == [] and pool.slots == []
