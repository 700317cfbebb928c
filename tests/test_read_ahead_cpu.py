"""read_ahead() of the pose drivers (host logic, no GPU): items in order, every index read exactly once, exceptions surface at the item
that raised, depth 0 = the plain loop."""
import threading

import pytest

from freepose_amd.scripts.dino_inference import read_ahead


class _Src:
    def __init__(self, fail_at=None):
        self.seen, self.threads, self.fail_at = [], set(), fail_at

    def __getitem__(self, i):
        self.seen.append(i)
        self.threads.add(threading.current_thread().name)
        if i == self.fail_at:
            raise KeyError(i)
        return i * i


@pytest.mark.parametrize("depth", [0, 1, 2, 5])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 17])
def test_read_ahead_order_and_coverage(depth, n):
    src = _Src()
    idx = list(range(3, 3 + n))
    assert list(read_ahead(src, idx, depth)) == [i * i for i in idx]
    assert src.seen == idx
    if depth > 0 and n >= 2:
        assert all(t.startswith("fp-frames") for t in src.threads)       # decoded off the caller's thread


def test_read_ahead_raises_where_the_loop_would():
    src = _Src(fail_at=5)
    got = []
    with pytest.raises(KeyError):
        for v in read_ahead(src, range(10), 2):
            got.append(v)
    assert got == [0, 1, 4, 9, 16]
