"""Round 6 (-m gpu): the whole `elprep sfm` path between two RANKS that are two PROCESSES - split phase
(sfm.route_device: elp_split_classify, elp_copy_records, elp_exchange_records), mark duplicates / metrics / BQSR count per split, THE
all-reduce of tables + counters, finalisation, ApplyBQSR, merge phase (sfm.emit_merged_device) - every output compared with the oracle run
split by split.  With a GPU per rank the records travel by ncclSend / ncclRecv and the tables by ncclAllReduce (RCCL over xGMI, through
the C ABI's device group); on a box with ONE GPU the same code runs with both ranks on it and the transport swapped (the group's
send-receive callback and torch.distributed over gloo).  And `bench.py --gpus 2` as the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc
from elprep_amd import sfm
from elprep_amd.batch import Batch
from tests import sfm_worker
from tests.test_sfm_cpu import _merge_reference
from tools import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _backend(world):
    import torch
    return "nccl" if torch.cuda.device_count() >= world else "gloo"


def _bam_fields(buf):
    """(refid, 1-based pos, flag, qname, qual bytes) of every record of a BAM record stream"""
    out, p = [], 0
    while p < buf.size:
        size = int(buf[p:p + 4].view(np.uint32)[0])
        rec = buf[p + 4:p + 4 + size]
        refid, pos = (int(x) for x in rec[0:8].view(np.int32))
        l_name, n_cig, flag, l_seq = int(rec[8]), int(rec[12:14].view(np.uint16)[0]), int(rec[14:16].view(np.uint16)[0]), int(rec[16:20].view(np.uint32)[0])
        name = rec[32:32 + l_name - 1].tobytes()
        q0 = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2
        out.append((refid, pos + 1, flag, name, rec[q0:q0 + l_seq].tobytes()))
        p += 4 + size
    return out


def test_sfm_two_ranks_as_two_processes_whole_path(tmp_path):
    world = 2
    backend = _backend(world)
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sfm_gpu_worker.py"), str(r), str(world), port, str(tmp_path), backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    res = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert bytes(res[0]["collective"]).decode() == ("cabi" if backend == "nccl" else "torch"), outs
    inputs = [sfm.unpack_batch(res[r]["input"]) for r in range(world)]
    cfg, gof, G, owner, _ = sfm_worker.make_rank_input(0, world, pairs_per_rank=2500)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    so = int(owner[G + 1])  # the spread split's owner
    # what arrives where, in arrival order: per routing round a rank's own records, then the other rank's
    halves = [[b.take(np.arange(b.n // 2)), b.take(np.arange(b.n // 2, b.n))] for b in inputs]
    want_local, want_spread = [[] for _ in range(world)], []
    for k in range(2):
        for r in range(world):
            for src in (r, 1 - r):
                bb = halves[src][k]
                g, sp = sfm.split_records(bb, gof)
                idx = np.nonzero(owner[g] == r)[0]
                if idx.size:
                    want_local[r].append(sfm.with_sr(bb, sp, g).take(idx))
                if r == so and sp.any():
                    want_spread.append(sfm.with_sr(bb, np.zeros(bb.n, bool), np.zeros(bb.n, np.uint16)).take(np.nonzero(sp)[0]))
    parts = {(r, 0): Batch.concat(want_local[r]) for r in range(world)}
    for r in range(world):
        parts[(r, 1)] = Batch.concat(want_spread) if r == so else sfm.empty_batch()
    assert parts[(so, 1)].n > 50 and sum(int(p.has_sr.sum()) for p in parts.values()) > 50
    # ---- the oracle: split file by split file (one `filter` run each, cmd/sfm.go), tables and counters summed
    oq = oc = ox = octr = None
    oflags, operms = {}, {}
    for key, p in parts.items():
        if p.n == 0:
            continue
        oflags[key] = np.zeros(p.n, np.uint16)
        order = []
        for sid in np.unique(p.split):
            sel = np.nonzero(p.split == sid)[0]
            sub = p.take(sel)
            perm = orc.sort_coordinate(sub, orc.mark_duplicates(sub, h))
            fl, c7, _ = orc.dup_metrics(sub, h, perm, 100)
            q, c, x = orc.bqsr_gather(sub, h, orc.BqsrRef(refs, sites), fl, 500)
            oflags[key][sel] = fl
            order.append((int(sid), sel[perm[:orc.num_sorted(sub)]]))
            oq, oc, ox, octr = (q, c, x, c7) if oq is None else (oq + q, oc + c, ox + x, octr + c7)
        operms[key] = np.concatenate([o for sid, o in sorted(order, key=lambda t: (t[0] == 0, t[0]))])
    fin = orc.BqsrFinal(oq, oc, ox, 500)
    oqual = {key: fin.apply(p, h, 0) for key, p in parts.items() if p.n}
    for r in range(world):
        assert np.array_equal(res[r]["qt"], oq) and np.array_equal(res[r]["ct"], oc) and np.array_equal(res[r]["xt"], ox), ("tables", r)
        assert np.array_equal(res[r]["ctr"], octr), ("counters", r)
        for w in (0, 1):
            p = parts[(r, w)]
            assert int(res[r][f"n{w}"][0]) == p.n, ("records that arrived", r, w)
            if p.n:
                assert np.array_equal(res[r][f"flags{w}"], oflags[(r, w)]), ("flags", r, w)
                assert np.array_equal(res[r][f"perm{w}"], operms[(r, w)]), ("order", r, w)
                assert np.array_equal(res[r][f"qual{w}"], oqual[(r, w)]), ("qualities", r, w)
    # ---- the merge phase: every rank's stream = its groups' output with the spread reads of those groups inserted (sam/split-merge.go:519-549)
    wsp = parts[(so, 1)]
    sorder = operms[(so, 1)]
    sgroup = gof[wsp.refid[sorder]]

    def fields(p, i, flags, qual):
        return (int(p.refid[i]), int(p.pos[i]), int(flags[i]), p.qname_of(int(i)), qual[int(p.qual_off[i]):int(p.qual_off[i + 1])].tobytes())
    n_spread_inserted = 0
    for r in range(world):
        loc, lorder = parts[(r, 0)], operms[(r, 0)]
        keys = [(int(loc.refid[i]), int(loc.pos[i])) for i in lorder]
        n_mapped = sum(1 for k in keys if k[0] >= 0)
        mine = np.nonzero(owner[sgroup] == r)[0]  # the spread reads of this rank's contig groups, in the spread file's order
        skeys = [(int(wsp.refid[sorder[j]]), int(wsp.pos[sorder[j]])) for j in mine]
        codes = _merge_reference(keys[:n_mapped], skeys)
        want = [fields(loc, lorder[c], oflags[(r, 0)], oqual[(r, 0)]) if c >= 0 else fields(wsp, sorder[mine[-c - 1]], oflags[(so, 1)], oqual[(so, 1)]) for c in codes]
        want += [fields(loc, i, oflags[(r, 0)], oqual[(r, 0)]) for i in lorder[n_mapped:]]
        got = _bam_fields(res[r]["merged"])
        assert len(got) == len(want), (r, len(got), len(want))
        first = next((k for k, (a, c) in enumerate(zip(got, want)) if a != c), None)
        assert first is None, (r, "first difference at", first, got[first][:4], want[first][:4])
        n_spread_inserted += len(mine)
    assert n_spread_inserted == len(sorder) > 50


@pytest.mark.fresh_only
def test_bench_with_two_ranks_as_the_driver_launches_it(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2: the N > 1 line (one rank per GPU over RCCL where there are two
    GPUs; ELP_BENCH_BACKEND=gloo with both ranks on the one GPU otherwise)"""
    world = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if _backend(world) == "gloo":
        env["ELP_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--reads", "600000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(line) == 1, p.stdout.decode()[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["workload"] and "roofline" in d


@pytest.mark.parametrize("ahead", [False, True])
@pytest.mark.parametrize("name,pairs,seed,pfrag,n_lanes", [("tiny", 6000, 3, 0.02, 4), ("tiny", 30000, 7, 0.0, 4), ("tiny", 2500, 11, 0.3, 20), ("tiny", 40, 13, 0.0, 1)])
def test_sort_metrics_and_the_bqsr_chain_at_once(name, pairs, seed, pfrag, n_lanes, ahead):
    """Behind elp_mark_duplicates the coordinate sort (side lane 1), the metrics pass (side lane 0) and gather -> finalize -> apply need
    nothing of each other: three host threads drive them at once on ONE context (what bench.py's default step and SfmRank.step do,
    include/elprep_hip.h at elp_sort_coordinate) - several rounds on the same context (snapshot / rollback), every output against the
    oracle each time.  ahead: elp_sort_ahead - the sort's key passes are queued from inside elp_mark_duplicates (rounds alternate with a
    plain sequential round, so a stale set of sorted words would show)."""
    from concurrent.futures import ThreadPoolExecutor
    from elprep_amd.engine import BqsrTables, Engine
    cfg = synth.config(name, seed)
    cfg.p_frag = pfrag
    cfg.n_lanes = n_lanes
    b = synth.generate(cfg, 0, pairs)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    oflags = orc.mark_duplicates(b, h)
    operm = orc.sort_coordinate(b, oflags)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0)
    e = Engine(h, 0)
    cuts = np.linspace(0, b.n, 4).astype(int)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.stage(b.take(np.arange(lo, hi)))
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    e.snapshot()
    e.sort_ahead(ahead)
    with ThreadPoolExecutor(1) as sort_pool, ThreadPoolExecutor(1) as mx_pool:
        for rnd in range(4):
            e.rollback()
            if ahead and rnd == 2:  # a sort WITHOUT mark duplicates in front (keys by the adapt stage, no passes made ahead), then on
                e.sort_ahead(False)
                assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(b))
                e.sort_ahead(True)
                e.rollback()
            e.mark_duplicates(True, fetch=False)
            st = sort_pool.submit(e.sort_coordinate)
            mx = mx_pool.submit(e.dup_metrics, 100)
            qt, ct, xt = e.recalibrate(500)
            lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
            qual = e.apply_bqsr(lut, present, 500)
            perm, ctr = st.result(), mx.result()
            assert np.array_equal(e.flags(), oflags), ("flags", rnd)
            assert np.array_equal(perm, operm), ("order", rnd)
            assert np.array_equal(ctr, octr), ("counters", rnd)
            assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox), ("tables", rnd)
            assert np.array_equal(qual, oqual), ("qualities", rnd)
    e.close()


# ---- the BGZF decoder in two phases (bgzf.hip, round 6): whatever zlib can write, byte for byte
def _records_with_payload_tags(raw, rec_off, seed):
    """every record of the stream gets an XB:B:C tag of bytes that DEFLATE treats differently: noise (long codes, stored blocks), runs
    (matches that overlap themselves), text, copies of data far back (distances up to 32 KB), nothing.  Returns the new stream + offsets."""
    import struct
    rng = np.random.default_rng(seed)
    src = raw.tobytes()
    far = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    out, offs = [], [0]
    total = 0
    for k in range(len(rec_off) - 1):
        rec = src[int(rec_off[k]):int(rec_off[k + 1])]
        kind = int(rng.integers(0, 8))
        n = int(rng.integers(1, 700))
        if k in (5, 9, 40):  # three records longer than a BGZF block: a run (nothing but matches of 258 at distance 1), noise (stored
            kind, n = (1, 0, 2)[(5, 9, 40).index(k)], 70000  # blocks), text (far copies of a period)
        if kind == 0:
            pay = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            pay = bytes([int(rng.integers(0, 256))]) * n
        elif kind == 2:
            pay = (b"ACGTTGCA" * (n // 8 + 1))[:n]
        elif kind == 3:
            at = int(rng.integers(0, len(far) - n))
            pay = far[at:at + n]
        elif kind == 4:
            pay = bytes(rng.choice(np.array([33, 35, 70, 70, 70, 58], dtype=np.uint8), n))
        elif kind == 5:
            pay = (bytes(rng.integers(0, 256, 3, dtype=np.uint8)) * 300)[:n]
        else:
            pay = b""
        tag = b"XBBC" + struct.pack("<I", len(pay)) + pay if kind != 7 else b""
        size = struct.unpack_from("<I", rec, 0)[0] + len(tag)
        new = struct.pack("<I", size) + rec[4:] + tag
        out.append(new)
        total += len(new)
        offs.append(total)
    return np.frombuffer(b"".join(out), dtype=np.uint8), np.asarray(offs, dtype=np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_stage_bgzf_inflates_whatever_zlib_wrote(seed):
    """random compression parameters per member (level 0-9, every strategy, memLevel 1-9, window 512 B-32 KB, member sizes 1 B-65280 B)
    over records with payload tags of every kind: the staged records give the sorted BAM bytes of elp_stage_bam on the inflated stream"""
    import struct
    import zlib
    from elprep_amd.engine import Engine
    from tests.test_gpu_round4 import _bam_case
    b, h, raw0, rec_off0 = _bam_case(2500, seed=30 + seed)
    raw, rec_off = _records_with_payload_tags(raw0, rec_off0, seed)
    stream = raw.tobytes()
    rng = np.random.default_rng(100 + seed)
    members, k = [], 0
    while k < len(stream):
        cut = int(rng.choice([1, 17, 300, 4099, 30011, 65280, 65280, 65280]))
        part = stream[k:k + cut]
        k += len(part)
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)), int(rng.integers(0, 5)))
        data = co.compress(part) + co.flush()
        members.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(data) + 25) + data +
                       struct.pack("<II", zlib.crc32(part), len(part)))
        if rng.random() < 0.05:
            members.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # an empty member in mid-stream
    bz = np.frombuffer(b"".join(members), dtype=np.uint8)
    outs = []
    for how in ("bam", "bgzf", "bgzf small pieces"):
        e = Engine(h, tuning={"bgzf_inflate_piece": 300_000, "bgzf_piece": 90_000, "bgzf_copy_chunk": 3} if how.endswith("pieces") else None)
        e.set_read_group_ids(h.rg_ids)
        if how == "bam":
            e.stage_bam(raw, rec_off=rec_off)
        else:
            e.stage_bgzf(bz)
        assert e.n == b.n
        e.mark_duplicates(True)
        e.sort_coordinate()
        outs.append(e.emit_sorted_bam().tobytes())
        if how == "bam":  # and the writer on the same records (runs, noise - stored blocks -, far copies): zlib reads what it wrote
            from tests.test_gpu_round4 import _members
            assert b"".join(m for _, m in _members(e.emit_sorted_bgzf().tobytes())) == outs[0]
        e.close()
    assert outs[0] == outs[1] and outs[0] == outs[2]


@pytest.mark.gpu
def test_stage_bgzf_survives_damaged_blocks():
    """bytes of the compressed data overwritten at random places: every call comes back with an error (a block that does not inflate, a
    CRC that does not match, records that do not chain) - or, where the damage fell on bits nothing reads, with the records; it never
    hangs and the context stays usable"""
    import zlib
    from elprep_amd.engine import ElpError, Engine
    from tests.test_gpu_round4 import _bam_case, _bgzf
    b, h, raw, rec_off = _bam_case(1500, seed=9)
    good = _bgzf(raw.tobytes(), 6)
    rng = np.random.default_rng(11)
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    errors = 0
    for trial in range(40):
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(18, len(bad) - 28))
            bad[at] = int(rng.integers(0, 256))
        e.reset()
        try:
            e.stage_bgzf(np.frombuffer(bytes(bad), dtype=np.uint8))
        except ElpError:
            errors += 1
            continue
        assert e.n == b.n  # (the damage changed nothing that matters - e.g. the byte was rewritten with its own value)
    assert errors >= 30, errors
    e.reset()
    e.stage_bgzf(np.frombuffer(good, dtype=np.uint8))
    assert e.n == b.n
    e.close()


@pytest.mark.gpu
def test_emit_sorted_bgzf_with_counts_that_want_codes_longer_than_15_bits():
    """records whose payload tags hold byte values with Fibonacci counts (the Huffman tree of such a block is deeper than 15: the length
    limit's repair, deflate_core.hpp limit_counts) and tags of two byte values only: zlib inflates every member to the record stream"""
    import struct
    import zlib
    from elprep_amd.engine import Engine
    from tests.test_gpu_round4 import _bam_case
    b, h, raw0, rec_off0 = _bam_case(1200, seed=77)
    rng = np.random.default_rng(5)
    fib = [1, 1]
    while sum(fib) < 52000:
        fib.append(fib[-1] + fib[-2])
    pool = np.frombuffer(b"".join(bytes([(37 * k + 11) % 256]) * f for k, f in enumerate(fib)), dtype=np.uint8)
    src = raw0.tobytes()
    out, offs, total = [], [0], 0
    for k in range(len(rec_off0) - 1):
        rec = src[int(rec_off0[k]):int(rec_off0[k + 1])]
        n = int(rng.integers(200, 900))
        pay = bytes(rng.choice(pool, n)) if k % 3 else bytes(rng.choice(np.array([0, 255], dtype=np.uint8), n))
        tag = b"XBBC" + struct.pack("<I", len(pay)) + pay
        new = struct.pack("<I", struct.unpack_from("<I", rec, 0)[0] + len(tag)) + rec[4:] + tag
        out.append(new)
        total += len(new)
        offs.append(total)
    raw = np.frombuffer(b"".join(out), dtype=np.uint8)
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    e.stage_bam(raw, rec_off=np.asarray(offs, dtype=np.uint64))
    e.mark_duplicates(True)
    e.sort_coordinate()
    want = e.emit_sorted_bam().tobytes()
    bz = e.emit_sorted_bgzf().tobytes()
    e.close()
    p, got, dynamic = 0, [], 0
    while p < len(bz):
        bsize = struct.unpack_from("<H", bz, p + 16)[0] + 1
        data = bz[p + 18:p + bsize - 8]
        dynamic += ((data[0] >> 1) & 3) == 2
        d = zlib.decompressobj(-15)
        part = d.decompress(data) + d.flush()
        crc, isize = struct.unpack_from("<II", bz, p + bsize - 8)
        assert d.eof and zlib.crc32(part) == crc and len(part) == isize
        got.append(part)
        p += bsize
    assert b"".join(got) == want and dynamic >= 3
    # and the other way round: the same stream as zlib compresses it (its own 15-bit-limited codes, Z_HUFFMAN_ONLY: literals only, the
    # longest codes there are) through the device's decoder
    from tests.test_gpu_round4 import _bgzf
    for level, strategy in ((9, 0), (6, 2)):
        e = Engine(h)
        e.set_read_group_ids(h.rg_ids)
        e.stage_bgzf(np.frombuffer(_bgzf(raw.tobytes(), level, strategy), dtype=np.uint8))
        assert e.n == b.n
        e.mark_duplicates(True)
        e.sort_coordinate()
        assert e.emit_sorted_bam().tobytes() == want
        e.close()
