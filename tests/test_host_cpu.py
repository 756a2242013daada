"""CPU tests (-m "not gpu"): the C-ABI libraries load and export every declared symbol; the host float code
(libelprep_host.so) agrees with the oracle; the synthetic generator is deterministic.  No GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
from elprep_amd import _lib
from elprep_amd.engine import BqsrTables, dup_derived, dup_metrics_report
from tests.common import dataset
from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)  # declarations, not the prose around them
    return sorted(set(re.findall(r"\b(elp_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.HIP_SO)  # must exist: build() makes it; no fallback
    names = sorted(set(_declared("elprep_hip.h")) | set(_declared("elprep_hip_debug.h")))  # the boundary + the harness header of the same library
    assert set(names) == set(_lib.HIP_SYMBOLS)
    assert set(_declared("elprep_hip_debug.h")) == {"elp_snapshot", "elp_rollback", "elp_set_tuning", "elp_profile_enable", "elp_profile_reset",
                                                    "elp_profile_count", "elp_profile_get", "elp_debug_check_guards"}
    for n in names:
        assert hasattr(L, n), n


def test_host_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.HOST_SO)
    names = _declared("elprep_host.h")
    assert set(names) == set(_lib.HOST_SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = _lib.hip().elp_create(0, C.byref(h))
    assert rc != 0 and not h.value


def test_synth_is_deterministic_and_shardable():
    cfg = synth.config("tiny")
    a = synth.generate(cfg, 0, 600)
    b1, b2 = synth.generate(cfg, 0, 250), synth.generate(cfg, 250, 600)
    from elprep_amd.batch import Batch
    c = Batch.concat([b1, b2])
    for name in ("refid", "pos", "flag", "mapq", "rgid", "qname", "cigar", "seq4", "qual", "qname_off", "qual_off", "tlen"):
        assert np.array_equal(getattr(a, name), getattr(c, name)), name
    # shape of the data: mates adjacent, duplicates and unmapped pairs present
    assert a.n >= 1200 and (a.flag & 0x4).any() and (a.flag & 0x800).any()


@pytest.fixture(scope="module")
def tables():
    cfg, b, h, refs, sites = dataset("tiny", 3000)
    flags = orc.mark_duplicates(b, h)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    return h, qt, ct, xt


def test_host_finalize_matches_oracle(tables):
    h, qt, ct, xt = tables
    fo = orc.BqsrFinal(qt, ct, xt, 500)
    ft = BqsrTables(qt, ct, xt, 500).finalize()
    for a, b in zip(fo.empirical(), ft.empirical()):
        assert np.array_equal(a, b)
    ro, eo, oo, mo, po = fo.combined()
    rt, et, ot, mt, pt = ft.combined()
    assert np.array_equal(ro, rt) and np.array_equal(eo, et) and np.array_equal(oo, ot) and np.array_equal(mo, mt) and np.array_equal(po, pt)
    for lv in (0, 4, 16):
        co, so = fo.quantize(lv)
        ct_, st = ft.quantize(lv)
        assert np.array_equal(co, ct_) and np.array_equal(so, st)


@pytest.mark.parametrize("levels,sqq", [(0, ()), (8, ()), (0, (10, 20, 30)), (0, (25,))])
def test_host_lut_matches_oracle_memo(tables, levels, sqq):
    """The factorised LUT must reproduce estimateHierarchicalBayesianQuality for every key the apply pass can hit."""
    h, qt, ct, xt = tables
    fo = orc.BqsrFinal(qt, ct, xt, 500)
    ft = BqsrTables(qt, ct, xt, 500).finalize()
    lut, present = ft.build_lut(levels, sqq)
    _, quantized = fo.quantize(levels)
    stat = orc.static_quantized_scores(sqq) if sqq else None
    rng = np.random.default_rng(1)
    quals = [q for q in range(94) if qt[:, q, 0].sum() > 0] + [6, 40, 93]
    checked = 0
    for cov in range(h.n_cov):
        assert present[cov] == 1
        for q in quals:
            for cyc in list(rng.integers(-150, 151, 12)) + [-500, 500, 1, -1]:
                if cyc == 0:
                    continue
                for cx in list(rng.integers(0, 16, 4)) + [-1]:
                    key = -1 if cx < 0 else (2 | ((cx & 3) << 4) | ((cx >> 2) << 6))
                    want = fo.recal_qual(cov, q, int(cyc), key, quantized, stat)
                    got = lut[cov, q, int(cyc) + 500, 16 if cx < 0 else ((key >> 4) & 15)]
                    assert got == want, (cov, q, cyc, cx)
                    checked += 1
    assert checked > 2000


def test_host_report_matches_oracle_text(tables):
    h, qt, ct, xt = tables
    fo = orc.BqsrFinal(qt, ct, xt, 500)
    ft = BqsrTables(qt, ct, xt, 500).finalize()
    assert ft.report(h.cov_names) == fo.report(h.cov_names)
    assert ft.report(h.cov_names, "ELP").startswith("#:ELPReport.v1.1:5\n")


def test_dup_derived_and_report():
    row = np.array([10, 1000, 5, 7, 2, 100, 20], dtype=np.int64)
    pct, ls = dup_derived(row)
    assert pct == pytest.approx((2 + 200) / (10 + 2000))
    assert ls == orc.estimate_library_size(1000 - 20, 1000 - 100)
    txt = dup_metrics_report(np.stack([row, np.zeros(7, np.int64)]), ["libA"], "elprep filter in out")
    assert "libA\t10\t1000\t5\t7\t2\t100\t20\t0.100498\t" in txt
    assert "Unknown Library\t0\t0\t0\t0\t0\t0\t0\tNaN\n" in txt


def test_dup_report_histogram_block():
    """'## HISTOGRAM' (filters/mark-optical-duplicates.go:628-697): only with exactly one library that has pairs; ROI column from
    estimateRoi, counts from the three set-size histograms, sizes beyond 100 as extra rows."""
    import math
    row = np.array([10, 1000, 5, 7, 2, 100, 20], dtype=np.int64)
    ctr = np.stack([row, np.zeros(7, np.int64)])
    hist = np.zeros((2, 3, 140), np.int64)
    hist[0, 0, 1], hist[0, 0, 2], hist[0, 0, 3], hist[0, 0, 120] = 800, 90, 9, 1   # all sets
    hist[0, 1, 1], hist[0, 1, 2], hist[0, 1, 119] = 815, 84, 1                     # non-optical
    hist[0, 2, 2], hist[0, 2, 3] = 18, 2                                            # optical
    txt = dup_metrics_report(ctr, ["libA"], "cmd", hist=hist)
    lines = txt.split("\n")
    k = lines.index("## HISTOGRAM\tjava.lang.Double")
    assert lines[k + 1] == "BIN\tCoverageMult\tall_sets\toptical_sets\tnon_optical_sets"
    _, ls = dup_derived(row)

    def fmt(f):
        s = "%.6f" % f
        t = s.rstrip("0")
        return s if t.endswith(".") else t

    for x in (1, 2, 3, 50, 100):
        roi = float(ls) * (1.0 - math.exp(-float(x * 1000) / float(ls))) / float(1000 - 100)
        assert lines[k + 1 + x] == "%d.0\t%s\t%d\t%d\t%d" % (x, fmt(roi), hist[0, 0, x], hist[0, 2, x], hist[0, 1, x]), x
    assert lines[k + 102] == "119.0\t0\t0\t0\t1" and lines[k + 103] == "120.0\t0\t1\t0\t0" and lines[k + 104] == ""
    # two libraries with pairs: no histogram block; without histograms: the old text
    two = np.stack([row, row, np.zeros(7, np.int64)])
    assert "## HISTOGRAM" not in dup_metrics_report(two, ["a", "b"], "cmd", hist=np.zeros((3, 3, 8), np.int64))
    assert dup_metrics_report(ctr, ["libA"], "cmd") == txt[:txt.index("## HISTOGRAM")] + "\n"


def test_host_pool_is_reentrant_and_arrays_can_be_reused(tables):
    """finalize / build_lut run their rows on a shared worker pool: calls from several host threads at once give the results of
    sequential calls, and filling a LUT pair of an earlier call again gives the same bytes as a fresh one."""
    import threading
    h, qt, ct, xt = tables
    ref = BqsrTables(qt, ct, xt, 500).finalize()
    lut0, present0 = ref.build_lut(0)
    emp0 = ref.empirical()
    results = [None] * 6

    def work(k):
        t = BqsrTables(qt * (1 if k % 2 == 0 else 2), ct * (1 if k % 2 == 0 else 2), xt * (1 if k % 2 == 0 else 2), 500).finalize()
        results[k] = (t.empirical(), t.build_lut(0))

    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    twice = BqsrTables(qt * 2, ct * 2, xt * 2, 500).finalize()
    lut2, present2 = twice.build_lut(0)
    for k in range(6):
        emp, (lut, present) = results[k]
        want_emp, want_lut, want_p = (emp0, lut0, present0) if k % 2 == 0 else (twice.empirical(), lut2, present2)
        assert all(np.array_equal(a, b) for a, b in zip(emp, want_emp))
        assert np.array_equal(lut, want_lut) and np.array_equal(present, want_p)
    # refill: first the doubled tables, then the original ones into the same arrays
    pair = twice.build_lut(0)
    again = ref.build_lut(0, out=pair)
    assert again[0] is pair[0] and np.array_equal(again[0], lut0) and np.array_equal(again[1], present0)


def test_tables_and_lut_in_rows_form_equal_the_dense_forms():
    """round 5: the host's table path on the rows of the qualities that occur (what crosses PCIe with many read groups) - the object built
    from rows finalizes to the same empirical qualities, and the LUT in rows form (rows + one default byte per other row) expands to
    the dense LUT byte for byte"""
    cfg, b, h, refs, sites = dataset("tiny", 6000, 3, 0.02)
    flags = orc.mark_duplicates(b, h)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    quals = [q for q in range(94) if qt[:, q, 0].any()] + [50]       # (a quality without entries may be among them)
    quals = sorted(set(quals))
    assert 3 <= len(quals) < 20
    dense = BqsrTables(qt, ct, xt, 500).finalize()
    rows = BqsrTables.from_rows(h.n_cov, quals, qt[:, quals], ct[:, quals], xt[:, quals], 500).finalize()
    for a, c in zip(dense.empirical(), rows.empirical()):
        assert np.array_equal(a, c)
    for levels, sqq in ((0, ()), (4, ()), (0, (10, 20, 30))):
        lut, present = dense.build_lut(levels, sqq)
        r, d, p = rows.build_lut_rows(quals, levels, sqq)
        assert np.array_equal(p, present)
        full = np.broadcast_to(d[:, :, None, None], lut.shape).copy()
        full[:, quals] = r
        assert np.array_equal(full, lut), (levels, sqq)
    with pytest.raises(RuntimeError):
        rows.build_lut_rows(quals[:-2])


def test_host_threads_setting_changes_nothing_but_the_pool():
    """ELP_HOST_THREADS (a host that runs one process per GPU divides its cores among them): FinalizeBQSRTables and the LUT are the same
    bytes with one worker as with the default pool - run in fresh processes, the pool is sized when the library is first used"""
    import subprocess
    import sys
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); from elprep_amd.engine import BqsrTables; rng = np.random.default_rng(5);"
            "n_cov, mc = 6, 40; ncyc = 2 * mc + 1;"
            "ct = np.zeros((n_cov, 94, ncyc, 2), np.int64); ct[:, 20:40, :, 0] = rng.integers(0, 50000, (n_cov, 20, ncyc)); ct[..., 1] = rng.binomial(ct[..., 0], 0.01);"
            "xt = np.zeros((n_cov, 94, 16, 2), np.int64); xt[:, 20:40, :, 0] = rng.integers(0, 900000, (n_cov, 20, 16)); xt[..., 1] = rng.binomial(xt[..., 0], 0.01);"
            "qt = np.zeros((n_cov, 94, 2), np.int64); qt[..., 0] = ct[..., 0].sum(2); qt[..., 1] = ct[..., 1].sum(2);"
            "tb = BqsrTables(qt, ct, xt, mc).finalize(); lut, present = tb.build_lut(0); q, c, x = tb.empirical();"
            "print(hashlib.sha256(lut.tobytes() + present.tobytes() + q.tobytes() + c.tobytes() + x.tobytes()).hexdigest())") % ROOT
    outs = []
    for threads in (None, "1", "3"):
        env = dict(os.environ)
        env.pop("ELP_HOST_THREADS", None)
        if threads:
            env["ELP_HOST_THREADS"] = threads
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip())
    assert len(outs[0]) == 64 and outs[0] == outs[1] == outs[2]
