"""CPU tests of the oracle's restatement of Go's math.Log / math.Lgamma (oracle/orc_gomath.c) and of the host library's own copy of
them (elprep_amd/host/bqsr_tables.cpp): the constants carry the bit patterns the Go sources print, the values are the correctly rounded
ones to within the algorithms' published error (< 1 ulp), and the host's FinalizeBQSRTables agrees with the oracle's on a million random
(observations, mismatches, reported quality) entries."""
import math
import struct

import numpy as np
import pytest

import oracle as orc
from elprep_amd.engine import BqsrTables


def _ulps(a: float, b: float) -> int:
    ia, ib = struct.unpack("<q", struct.pack("<d", a))[0], struct.unpack("<q", struct.pack("<d", b))[0]
    return abs(ia - ib)


def test_constants_have_the_bit_patterns_of_the_go_source():
    assert orc.gomath_selfcheck() == 0


def test_go_log_is_within_one_ulp_of_the_correctly_rounded_value():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(3)
    xs = np.concatenate([np.exp(rng.uniform(-60, 60, 4000)), rng.uniform(0.5, 2.0, 4000), [1.0, 2.0, 10.0, 0.5, 1e-300, 1e300]])
    differs = 0
    for x in xs:
        x = float(x)
        got, want = orc.go_log(x), float(mp.log(mp.mpf(x)))
        assert _ulps(got, want) <= 1, (x, got, want)
        differs += got != math.log(x)
    assert orc.go_log(1.0) == 0.0 and orc.go_log(2.0) == 0.6931471805599453
    assert differs > 0  # Go's log is not glibc's: that is the reason the restatement exists


def test_go_exp_is_the_pure_go_function():
    """math/exp.go (FreeBSD's e_exp.c): < 1 ulp from the correctly rounded value, the documented special cases, the same bits in the
    oracle's C and the host library's C++ copy (through math.Pow(10, .), the one caller the two share an entry point for) - and NOT glibc's
    exp everywhere, which is why it is restated (round 6; the amd64 assembly caveat is in oracle/orc_gomath.c)."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-700, 700, 3000), rng.uniform(-1, 1, 3000), rng.uniform(-1e-3, 1e-3, 500), [0.0, 1.0, -1.0, 0.5, 709.0, -745.0]])
    differs = 0
    for x in xs:
        x = float(x)
        got, want = orc.go_exp(x), float(mp.exp(mp.mpf(x)))
        assert _ulps(got, want) <= 1, (x, got, want)
        differs += got != math.exp(x)
    assert orc.go_exp(0.0) == 1.0 and orc.go_exp(1e-10) == 1.0 + 1e-10                      # |x| < 2^-28: 1 + x
    assert orc.go_exp(710.0) == math.inf and orc.go_exp(-746.0) == 0.0                        # beyond Overflow / Underflow
    assert orc.go_exp(math.inf) == math.inf and orc.go_exp(-math.inf) == 0.0 and math.isnan(orc.go_exp(math.nan))
    assert orc.go_exp(1.0) == 2.718281828459045
    assert differs > 0
    # the host library's own copy of the function (elprep_amd/host/bqsr_tables.cpp: go_exp) through its one caller with an entry point of
    # its own, the library-size estimate (filters/mark-optical-duplicates.go:537-569: a bisection on exp(-n / x)), bit for bit
    from elprep_amd.engine import dup_derived
    for _ in range(300):
        pairs = int(rng.integers(2, 5_000_000))
        dups = int(rng.integers(1, pairs))
        opt = int(rng.integers(0, dups + 1))
        row = np.array([0, pairs, 0, 0, 0, dups, opt], np.int64)
        assert dup_derived(row)[1] == orc.estimate_library_size(pairs - opt, pairs - dups), (pairs, dups, opt)


def test_go_lgamma_on_counts_and_in_every_branch():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    assert orc.go_lgamma(1.0) == 0.0 and orc.go_lgamma(2.0) == 0.0
    # 3 .. 7: the logarithm of the product the source builds ((y + 2) .. (y + 6) with y = 0)
    for n, fact in ((3, 2.0), (4, 6.0), (5, 24.0), (6, 120.0), (7, 720.0)):
        assert orc.go_lgamma(float(n)) == orc.go_log(fact)
    rng = np.random.default_rng(4)
    xs = np.concatenate([np.arange(1, 3000, dtype=float), np.floor(np.exp(rng.uniform(0, 21, 3000))), rng.uniform(0.01, 9.0, 3000)])
    for x in xs:
        x = float(x)
        if 0.19 < x < 0.2317 or 1.19 < x < 1.2317:
            continue  # the two windows where the Go source's interval bounds differ from FreeBSD's (see orc_gomath.c): not on any caller's path
        want = mp.loggamma(mp.mpf(x))
        err = abs(mp.mpf(orc.go_lgamma(x)) - want)
        # the Sun algorithm's error bound: a few units of the last place of max(|lgamma|, the terms that cancel near 1 and 2)
        assert err <= 4e-16 * max(abs(want), 1) + 1e-16, (x, float(err))


def test_host_finalize_agrees_with_the_oracle_on_a_million_entries():
    """8 covariates x 94 qualities x 1401 cycles = 1.05 M (observations, mismatches) pairs, log-uniform up to 2^31 and beyond (the
    clamp of calculateBayesianEstimateOfEmpiricalQuality, bqsr.go:623-628): every EmpiricalQuality of the host library (hoisted Lgamma
    terms, worker pool) equals the oracle's direct restatement"""
    rng = np.random.default_rng(5)
    n_cov, mc = 8, 700
    ncyc = 2 * mc + 1
    obs = np.floor(np.exp(rng.uniform(0, 22.5, (n_cov, 94, ncyc)))).astype(np.int64)
    obs[rng.random(obs.shape) < 0.02] = 0
    obs[rng.random(obs.shape) < 0.001] = 3_000_000_000
    rate = np.exp(rng.uniform(-9, -0.5, obs.shape))
    mism = np.minimum(obs, rng.binomial(np.minimum(obs, 2**31 - 1), rate).astype(np.int64))
    ct = np.stack([obs, mism], axis=-1)
    qt = ct.sum(axis=2)
    xt = np.zeros((n_cov, 94, 16, 2), np.int64)
    xt[:, :, 0] = qt
    fo = orc.BqsrFinal(qt, ct, xt, mc)
    ft = BqsrTables(qt, ct, xt, mc).finalize()
    for a, b in zip(fo.empirical(), ft.empirical()):
        assert np.array_equal(a, b)
    assert int((ct[..., 0] > 0).sum()) > 1_000_000


def test_host_lut_agrees_with_the_oracle_on_random_tables():
    """The host evaluates every table entry under two priors in one pass (its reported quality for FinalizeBQSRTables, the row's
    conditional estimate for the LUT) and keeps the cycle / context tables as lazily zeroed arrays with an index of the rows that hold
    anything.  Random tables - dense rows, rows that hold a single entry, rows whose quality entry is empty although cycle entries are
    not (no gather produces those; the index must come from the rows themselves) - and 30 000 random LUT keys against the oracle's
    direct estimateHierarchicalBayesianQuality (bqsr.go:901-919)."""
    rng = np.random.default_rng(17)
    n_cov, mc = 3, 300
    ncyc = 2 * mc + 1
    obs = np.floor(np.exp(rng.uniform(0, 21, (n_cov, 94, ncyc)))).astype(np.int64)
    obs[:, rng.random(94) < 0.5] = 0                       # half of the qualities do not occur at all
    obs[rng.random(obs.shape) < 0.3] = 0
    lonely = rng.integers(0, 94, 6)
    obs[:, lonely] = 0
    obs[:, lonely, rng.integers(0, ncyc, 6)] = 12345       # rows with one entry
    mism = rng.binomial(np.minimum(obs, 2**31 - 1), np.exp(rng.uniform(-9, -0.5, obs.shape))).astype(np.int64)
    ct = np.stack([obs, mism], axis=-1)
    qt = ct.sum(axis=2)
    xobs = np.floor(np.exp(rng.uniform(0, 21, (n_cov, 94, 16)))).astype(np.int64) * (qt[..., 0:1] > 0)
    xt = np.stack([xobs, rng.binomial(np.minimum(xobs, 2**31 - 1), 0.01).astype(np.int64)], axis=-1)
    orphan = int(lonely[0])
    qt[:, orphan] = 0                                       # quality entry empty, cycle entry not
    fo = orc.BqsrFinal(qt, ct, xt, mc)
    ft = BqsrTables(qt, ct, xt, mc).finalize()
    for a, b in zip(fo.empirical(), ft.empirical()):
        assert np.array_equal(a, b)
    lut, present = ft.build_lut(0)
    _, quantized = fo.quantize(0)
    assert present.tolist() == [1] * n_cov
    for _ in range(30000):
        cov, q, cyc, cx = int(rng.integers(0, n_cov)), int(rng.integers(0, 94)), int(rng.integers(-mc, mc + 1)), int(rng.integers(-1, 16))
        if cyc == 0:
            continue
        key = -1 if cx < 0 else (2 | ((cx & 3) << 4) | ((cx >> 2) << 6))
        assert lut[cov, q, cyc + mc, 16 if cx < 0 else cx] == fo.recal_qual(cov, q, cyc, key, quantized, None), (cov, q, cyc, cx)
