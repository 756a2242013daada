"""CPU tests: the all-cores variants of the oracle (bench.py's CPU baseline: parallel merge sort, sharded duplicate-marking maps,
thread-private BQSR tables) give the sequential oracle's results at any thread count."""
import numpy as np
import pytest

import oracle as orc
from tests.common import dataset


@pytest.mark.parametrize("nt", [1, 3, 8])
def test_multithreaded_oracle_equals_sequential(nt):
    cfg, b, h, refs, sites = dataset("tiny", 4000, 1, 0.05)
    f1 = orc.mark_duplicates(b, h)
    p1 = orc.sort_coordinate(b, f1)
    fl, c1, _ = orc.dup_metrics(b, h, p1, 100)
    f2, _ = orc.dup_metrics_mt(b, h, None, 100, nt)
    assert np.array_equal(f2, f1)
    p2 = orc.sort_coordinate_mt(b, f2, nt)
    assert np.array_equal(p1, p2)
    f3, c2 = orc.dup_metrics_mt(b, h, p2, 100, nt)
    assert np.array_equal(fl, f3) and np.array_equal(c1, c2)
    ref = orc.BqsrRef(refs, sites)
    q1 = orc.bqsr_gather(b, h, ref, fl, 500)
    q2 = orc.bqsr_gather_mt(b, h, ref, fl, 500, nt)
    assert all(np.array_equal(a, x) for a, x in zip(q1, q2))
    fin = orc.BqsrFinal(*q1, 500)
    assert np.array_equal(fin.apply(b, h, 0), orc.bqsr_apply_mt(fin, b, h, 0, (), nt))
