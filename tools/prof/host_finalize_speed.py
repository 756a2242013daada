"""Wall-clock of the host's table path (elprep_amd/host: tables object, FinalizeBQSRTables, LUT) by table shape, on the cores of the
box it runs on.  Usage: python tools/prof/host_finalize_speed.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from elprep_amd.engine import BqsrTables  # noqa: E402

MAXC, NQ, NX = 500, 94, 16


def make(nq, ncov, seed=1):
    rng = np.random.default_rng(seed)
    ncyc = 2 * MAXC + 1
    qt = np.zeros((ncov, NQ, 2), np.int64)
    ct = np.zeros((ncov, NQ, ncyc, 2), np.int64)
    xt = np.zeros((ncov, NQ, NX, 2), np.int64)
    quals = [2, 11, 25, 37][:nq] if nq <= 4 else [2] + list(range(3, 3 + nq - 1))
    cyc = list(range(MAXC - 150, MAXC)) + list(range(MAXC + 1, MAXC + 151))
    for q in quals:
        obs = rng.integers(100000, 3000000, size=(ncov, len(cyc)))
        mm = (obs * 10 ** (-q / 10) * rng.uniform(0.5, 2, size=obs.shape)).astype(np.int64)
        ct[:, q, cyc, 0] = obs
        ct[:, q, cyc, 1] = mm
        qt[:, q, 0] = obs.sum(1)
        qt[:, q, 1] = mm.sum(1)
        xo = rng.integers(1000000, 30000000, size=(ncov, NX))
        xt[:, q, :, 0] = xo
        xt[:, q, :, 1] = (xo * 10 ** (-q / 10)).astype(np.int64)
    return qt, ct, xt


print("cpus", len(os.sched_getaffinity(0)))
for ncov, nq in ((1, 4), (1, 40), (4, 40), (16, 4), (16, 8)):
    qt, ct, xt = make(nq, ncov)
    lutbuf = None
    best = [1e9] * 3
    for rep in range(6):
        t0 = time.perf_counter()
        tb = BqsrTables(qt, ct, xt, MAXC)
        t1 = time.perf_counter()
        tb.finalize()
        t2 = time.perf_counter()
        lutbuf = tb.build_lut(0, out=lutbuf)
        t3 = time.perf_counter()
        for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2)):
            best[k] = min(best[k], v)
    print("%2d covariates x %2d qualities: tables %.2f  finalize %.2f  lut %.2f ms" % ((ncov, nq) + tuple(v * 1e3 for v in best)))
