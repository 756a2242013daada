"""GPU parity test (-m gpu) of the sfm layer on ONE GPU: the splits of two ranks are processed rank after rank (the all-reduce
is a plain sum here; the world_size-2 exchange itself is covered on CPU over gloo, tests/test_sfm_cpu.py) and every output is
compared with the oracle run split by split — parity is per command: this is `elprep sfm`, not `filter`."""
import numpy as np
import pytest

import oracle as orc
from elprep_amd import sfm
from elprep_amd.batch import Batch
from elprep_amd.engine import BqsrTables
from tests import sfm_worker
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 1])
def test_sfm_ranks_on_one_gpu(world):
    """world = 1: both contig groups, the unmapped split (one context, three split ids) and the spread split on one rank"""
    inputs, owner, gof, G, cfg = [], None, None, None, None
    for r in range(world):
        cfg, gof, G, owner, b = sfm_worker.make_rank_input(r, world, pairs_per_rank=4000)
        inputs.append(b)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    # what the routing delivers to each rank (same construction as the gloo test checks against the real exchange)
    local = [[] for _ in range(world)]
    spread = []
    for b in inputs:
        g, sp = sfm.split_records(b, gof)
        tagged = sfm.with_sr(b, sp, g)
        for r in range(world):
            idx = np.nonzero(owner[g] == r)[0]
            if idx.size:
                local[r].append(tagged.take(idx))
        if sp.any():
            spread.append(b.take(np.nonzero(sp)[0]))
    parts = {}
    for r in range(world):
        parts[(r, 0)] = Batch.concat(local[r])
        parts[(r, 1)] = Batch.concat(spread) if r == owner[G + 1] else sfm.empty_batch()
    assert parts[(int(owner[G + 1]), 1)].n > 50 and sum(int(p.has_sr.sum()) for p in parts.values()) > 50

    # ---- device: rank after rank
    ranks = []
    tot = None
    for r in range(world):
        rk = sfm.SfmRank(h, 0, sfm.Comm())
        rk.stage(0, parts[(r, 0)])
        rk.stage(1, parts[(r, 1)])
        for k in range(h.n_ref):
            rk.set_reference(k, refs[k])
            rk.set_known_sites(k, sites[k])
        out = rk.gather(500, 100)
        flat = np.concatenate([a.ravel() for a in out])
        tot = flat if tot is None else tot + flat  # the all-reduce
        shapes = [a.shape for a in out]
        ranks.append(rk)
    tabs, at = [], 0
    for shp in shapes:
        n = int(np.prod(shp))
        tabs.append(tot[at:at + n].reshape(shp))
        at += n
    qt, ct, xt, ctr = tabs

    # ---- oracle: split file by split file (one `filter` run each, as cmd/sfm.go runs them)
    oq = oc = ox = octr = None
    oflags, operms = {}, {}
    for key, p in parts.items():
        if p.n == 0:
            continue
        oflags[key] = np.zeros(p.n, np.uint16)
        out_order = []
        for sid in np.unique(p.split):
            sel = np.nonzero(p.split == sid)[0]
            sub = p.take(sel)
            perm = orc.sort_coordinate(sub, orc.mark_duplicates(sub, h))
            fl, c7, _ = orc.dup_metrics(sub, h, perm, 100)
            q, c, x = orc.bqsr_gather(sub, h, orc.BqsrRef(refs, sites), fl, 500)
            oflags[key][sel] = fl
            out_order.append((int(sid), sel[perm[:orc.num_sorted(sub)]]))
            oq = q if oq is None else oq + q
            oc = c if oc is None else oc + c
            ox = x if ox is None else ox + x
            octr = c7 if octr is None else octr + c7
        # the context's output: its contig groups in @SQ order, then the unmapped split (id 0)
        operms[key] = np.concatenate([o for sid, o in sorted(out_order, key=lambda t: (t[0] == 0, t[0]))])
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    assert np.array_equal(ctr, octr)
    for (r, w), fl in oflags.items():
        eng = ranks[r].engines[w]
        assert np.array_equal(eng.flags(), fl), (r, w)
        assert np.array_equal(eng.permutation()[:eng.n_sorted], operms[(r, w)]), (r, w)
    # the tagged copies never reach the tables: the same reads without the copies give the same BQSR tables
    tb = BqsrTables(qt, ct, xt, 500).finalize()
    lut, present = tb.build_lut(0)
    fin = orc.BqsrFinal(oq, oc, ox, 500)
    for r in range(world):
        ranks[r].apply(lut, present, 500)
        for w in (0, 1):
            p = parts[(r, w)]
            if p.n:
                assert np.array_equal(ranks[r].engines[w].qual(), fin.apply(p, h, 0)), (r, w)
        ranks[r].close()


@pytest.mark.parametrize("form", ["dense", "rows"])
def test_sfm_one_rank_step_with_the_host_behind_the_sorts(form):
    """SfmRank.step on ONE rank through the C ABI's device group (a group of one): the same flags, permutation, tables, counters and
    qualities as the oracle run split by split - the order of events differs from gather() + apply() (metrics in front of the sort, the
    finalisation on a host thread while the GPU sorts), the results must not.  "rows": the tables and the LUT travel in rows form (round 5)"""
    from concurrent.futures import ThreadPoolExecutor
    cfg, gof, G, owner, b = sfm_worker.make_rank_input(0, 1, pairs_per_rank=6000)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    g, sp = sfm.split_records(b, gof)
    parts = [sfm.with_sr(b, sp, g), b.take(np.nonzero(sp)[0])]
    rk = sfm.SfmRank(h, 0, sfm.Comm(), collective="cabi")
    assert rk.collective == "none"  # a group of one: the device path without a communicator
    for w in (0, 1):
        rk.stage(w, parts[w])
    for k in range(h.n_ref):
        rk.set_reference(k, refs[k])
        rk.set_known_sites(k, sites[k])
    box = {}

    def finalize(qt, ct, xt):
        box["tables"] = (qt.copy(), ct.copy(), xt.copy())
        return BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    def finalize_rows(quals, q_rows, c_rows, x_rows):
        tb = BqsrTables.from_rows(h.n_cov, quals, q_rows, c_rows, x_rows, 500).finalize()
        box["rows"] = (list(quals), q_rows.copy(), c_rows.copy(), x_rows.copy())
        return tb.build_lut_rows(quals, 0)
    with ThreadPoolExecutor(1) as pool:
        ctr = rk.step(500, 100, pool, finalize, finalize_rows if form == "rows" else None)
    rk.sync()
    if form == "rows":  # the rows that came back, put where they belong in the dense tables
        assert "tables" not in box, "the rows form was not taken"
        quals, qr, cr, xr = box["rows"]
        qt = np.zeros((h.n_cov, 94, 2), np.int64); ct = np.zeros((h.n_cov, 94, 1001, 2), np.int64); xt = np.zeros((h.n_cov, 94, 16, 2), np.int64)
        qt[:, quals], ct[:, quals], xt[:, quals] = qr.reshape(h.n_cov, len(quals), 2), cr.reshape(h.n_cov, len(quals), 1001, 2), xr.reshape(h.n_cov, len(quals), 16, 2)
        box["tables"] = (qt, ct, xt)
    oq = oc = ox = octr = None
    for w, p in enumerate(parts):
        fl_all = np.zeros(p.n, np.uint16)
        for sid in np.unique(p.split):
            sel = np.nonzero(p.split == sid)[0]
            sub = p.take(sel)
            perm = orc.sort_coordinate(sub, orc.mark_duplicates(sub, h))
            fl, c7, _ = orc.dup_metrics(sub, h, perm, 100)
            q, c, x = orc.bqsr_gather(sub, h, orc.BqsrRef(refs, sites), fl, 500)
            fl_all[sel] = fl
            oq = q if oq is None else oq + q
            oc = c if oc is None else oc + c
            ox = x if ox is None else ox + x
            octr = c7 if octr is None else octr + c7
        assert np.array_equal(rk.engines[w].flags(), fl_all), w
    qt, ct, xt = box["tables"]
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    assert np.array_equal(ctr, octr)
    fin = orc.BqsrFinal(oq, oc, ox, 500)
    for w, p in enumerate(parts):
        assert np.array_equal(rk.engines[w].qual(), fin.apply(p, h, 0)), w
    rk.close()
