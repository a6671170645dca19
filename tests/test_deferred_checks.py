"""CPU: the deferred device-side checks of eprecon_amd._lib (values whose verification rides on a later blocking count read):
every read site drains the entries of ITS stream in the same transfer, entries of another stream stay, a mismatch raises, the
list is bounded and guarded by a lock (the pipelined panoptic worker reads from a second thread)."""
import threading

import pytest

torch = pytest.importorskip("torch")
from eprecon_amd import _lib  # noqa: E402


@pytest.fixture(autouse=True)
def fake_stream(monkeypatch):
    state = {"stream": 11}
    monkeypatch.setattr(_lib, "current_stream", lambda: state["stream"])
    _lib.take_deferred(None)
    yield state
    _lib.take_deferred(None)


def test_read_counts_takes_the_pending_checks_of_its_stream_along(fake_stream):
    ok = torch.tensor([7], dtype=torch.int32)
    _lib.defer_check(ok, 7, "fine")
    fake_stream["stream"] = 22
    _lib.defer_check(torch.tensor([1], dtype=torch.int32), 0, "other stream")
    fake_stream["stream"] = 11
    before = _lib.HOST_READS
    assert _lib.read_counts(torch.tensor([[3, 4], [5, 6]], dtype=torch.int32)) == [3, 4, 5, 6]
    assert _lib.HOST_READS == before + 1
    assert [it[2] for it in _lib._DEFERRED] == ["other stream"]          # left for a read on stream 22
    fake_stream["stream"] = 22
    with pytest.raises(_lib.EpreconError, match="other stream: expected 0, the device reports 1"):
        _lib.read_counts(torch.zeros(2, dtype=torch.int32))
    assert _lib._DEFERRED == []


def test_a_mismatch_raises_at_the_next_read_and_is_not_reported_twice():
    _lib.defer_check(torch.tensor([5], dtype=torch.int32), 9, "rows with a batch index out of range")
    with pytest.raises(_lib.EpreconError, match="batch index out of range"):
        _lib.read_counts(torch.zeros(1, dtype=torch.int32))
    assert _lib.read_counts(torch.ones(1, dtype=torch.int32)) == [1]


def test_drain_costs_a_read_only_when_something_is_pending():
    before = _lib.HOST_READS
    _lib.drain_deferred()
    assert _lib.HOST_READS == before
    _lib.defer_check(torch.tensor([0], dtype=torch.int32), 0, "x")
    _lib.drain_deferred()
    assert _lib.HOST_READS == before + 1 and _lib._DEFERRED == []


def test_list_is_bounded_and_thread_safe():
    def work():
        for k in range(200):
            _lib.defer_check(torch.tensor([k], dtype=torch.int32), k, "t")
    threads = [threading.Thread(target=work) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(_lib._DEFERRED) == _lib._DEFERRED_MAX
    _lib.drain_deferred()       # all self-consistent: no raise


def test_overflow_verifies_the_evicted_entry_instead_of_dropping_it():
    """more than _DEFERRED_MAX entries pending on a stream that never reads -> the oldest is CHECKED when it leaves"""
    _lib.defer_check(torch.tensor([3], dtype=torch.int32), 4, "oldest entry")
    for k in range(_lib._DEFERRED_MAX - 1):
        _lib.defer_check(torch.tensor([k], dtype=torch.int32), k, "t")
    before = _lib.HOST_READS
    with pytest.raises(_lib.EpreconError, match="oldest entry: expected 4, the device reports 3"):
        _lib.defer_check(torch.tensor([0], dtype=torch.int32), 0, "one too many")
    assert _lib.HOST_READS == before + 1 and len(_lib._DEFERRED) == _lib._DEFERRED_MAX


def test_abandoned_pinned_read_hands_its_checks_back(monkeypatch):
    class FakeRead(_lib.PinnedRead):      # (the real constructor needs a device: only the bookkeeping is under test)
        def __init__(self):
            self._pending = _lib.take_deferred()
    _lib.defer_check(torch.tensor([1], dtype=torch.int32), 2, "taken along, never verified")
    r = FakeRead()
    assert _lib._DEFERRED == []
    del r
    with pytest.raises(_lib.EpreconError, match="never verified"):
        _lib.drain_deferred()
