import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fresh_only: GPU test that must not run on a reused context (it tests what a NEW context does: missing "
                                       "inputs, creation, destruction) - every other GPU test runs twice, [fresh] and [reused]")


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must not silently pass: skip with a reason instead.
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------------------------------------------------------------------------
# Context reuse as a test dimension (round 6).  The one wrong-answer bug of five rounds lived in a context that was
# REUSED behind a larger read set (scratch slots that had grown before: profiles/round5_scratch_aliasing_found_and_fixed.txt), and every
# parity test made a fresh context per case.  Every `-m gpu` test now runs twice:
#   [fresh]   as written: `Engine(header)` creates a context, `close()` destroys it;
#   [reused]  `Engine(header)` hands out a POOLED context of the same header instead - one that already ran the whole path on a larger
#             read set (mark duplicates, sort, metrics, gather, apply: every scratch slot, column and side buffer has grown and holds
#             stale bytes) and every earlier [reused] test with that header - after `elp_reset`; `close()` gives it back to the pool.
# The reference consumes and reuses a `Sam` across pipelines the same way (sam/filter-pipeline.go:242-246).
# ELP_DEBUG_POISON=<byte> in the environment adds pre-filled device buffers to either mode (tools/prof/final_round6.sh runs both).
# ---------------------------------------------------------------------------------------------------------------------------------
_POOL = {}          # header key -> [free pooled engines]
_OUT = []           # pooled engines handed out during the current test
_STATE = {"reuse": False, "primed": 0, "reused": 0}
_TUNE_DEFAULTS = {"count_kernel": 0, "apply_kernel": 0, "bgzf_piece": 1 << 30, "bgzf_weak_guess": 0, "score_kernel": 0, "count3_rlog": -1, "qual_hint": 0,
                  "qual_hint_drop": -1, "pair_table_slots": 0, "mate_path": 0, "tie_rounds": 0, "radix_tile": 0, "sort_pairs": 0, "exchange_piece": 0,
                  "bgzf_stored": 0, "md_fused": 0, "apply_wgs": 0, "presort_tile": 0, "side_priority": 0, "bgzf_inflate": 0, "bgzf_fixed": 0, "bgzf_copy_chunk": 0, "bgzf_first_chunk_div": 4, "bgzf_tok_lds": 0, "bgzf_tok_fail_above": 0, "bgzf_inflate_piece": 2 << 30}


def _header_key(header, device, flat_abi):
    return (device, bool(flat_abi), header.ref_len.tobytes(), header.rg_lib.tobytes(), header.rg_cov.tobytes(), header.n_lib, header.n_cov)


def _priming_batch(header, n_pairs=3000, read_len=100, seed=12345):
    """A read set larger than most test inputs that fits ANY header: paired reads in aligner order on the header's contigs, a few
    duplicates, fragments, unmapped mates and soft clips - what makes every stage allocate and fill its buffers."""
    import numpy as np
    from elprep_amd.batch import Batch
    rng = np.random.default_rng(seed)
    n = 2 * n_pairs
    n_ref, n_rg = header.n_ref, header.n_rg
    L = read_len
    refid = np.full(n, -1, np.int32)
    pos = np.zeros(n, np.int32)
    flag = np.zeros(n, np.uint16)
    tlen = np.zeros(n, np.int32)
    cig_len = np.zeros(n, np.int64)
    cigs = []
    for k in range(n_pairs):
        a, b = 2 * k, 2 * k + 1
        if n_ref == 0 or rng.random() < 0.02:  # unmapped pair
            flag[a], flag[b] = 0x1 | 0x4 | 0x8 | 0x40, 0x1 | 0x4 | 0x8 | 0x80
            cigs += [[], []]
            continue
        r = int(rng.integers(0, n_ref))
        room = max(int(header.ref_len[r]) - 2 * L - 60, 1)
        p = int(rng.integers(1, room + 1)) if rng.random() > 0.15 else 1 + (k % 7)  # (a pile-up at the contig's start: duplicates, long tie runs)
        q = min(p + int(rng.integers(0, 50)) + L, max(int(header.ref_len[r]) - L, 1))
        refid[a] = refid[b] = r
        pos[a], pos[b] = p, q
        flag[a], flag[b] = 0x1 | 0x2 | 0x20 | 0x40, 0x1 | 0x2 | 0x10 | 0x80
        tlen[a], tlen[b] = q + L - p, -(q + L - p)
        for x in (a, b):
            if rng.random() < 0.1:
                s = int(rng.integers(1, 9))
                cigs.append([(s << 4) | 4, ((L - s) << 4) | 0])
            elif rng.random() < 0.1:
                m1 = int(rng.integers(10, L - 12))
                cigs.append([(m1 << 4) | 0, (2 << 4) | 1, ((L - m1 - 2) << 4) | 0])
            else:
                cigs.append([(L << 4) | 0])
        if rng.random() < 0.03:  # the mate is unmapped: a true fragment
            flag[a] = 0x1 | 0x8 | 0x40
            flag[b] = 0x1 | 0x4 | 0x80
            cigs[b] = []
    for i, c in enumerate(cigs):
        cig_len[i] = len(c)
    next_refid = refid.reshape(-1, 2)[:, ::-1].reshape(-1).copy()
    pnext = pos.reshape(-1, 2)[:, ::-1].reshape(-1).copy()
    names = [("P%d:%d:%d" % (k % 3, 1000 + (k * 37) % 900, k)).encode() for k in range(n_pairs)]
    qn = np.frombuffer(b"".join(nm + nm for nm in names), dtype=np.uint8).copy()
    ql = np.repeat(np.array([len(nm) for nm in names], dtype=np.uint64), 2)
    sb = (L + 1) // 2
    return Batch(refid=refid, pos=pos, next_refid=next_refid, pnext=pnext, tlen=tlen, flag=flag,
                 mapq=rng.integers(1, 61, n).astype(np.uint8), rgid=(rng.integers(0, n_rg, n).astype(np.uint16) if n_rg else np.full(n, 0xFFFF, np.uint16)),
                 has_sr=np.zeros(n, np.uint8), l_seq=np.full(n, L, np.uint32),
                 qname_off=np.concatenate([[0], np.cumsum(ql)]).astype(np.uint64), qname=qn,
                 cigar_off=np.concatenate([[0], np.cumsum(cig_len)]).astype(np.uint64),
                 cigar=np.asarray([o for c in cigs for o in c], dtype=np.uint32),
                 seq_off=(np.arange(n + 1, dtype=np.uint64) * sb), seq4=rng.choice(np.array([0x11, 0x12, 0x24, 0x48, 0x81, 0x88, 0x42, 0x1F], np.uint8), n * sb),
                 qual_off=(np.arange(n + 1, dtype=np.uint64) * L), qual=rng.choice(np.array([2, 6, 11, 22, 30, 37, 40], np.uint8), n * L),
                 split=np.zeros(n, np.uint16))


def _prime(engine):
    """the whole path once on the priming read set: the context the test gets has grown buffers full of another read set's bytes"""
    import numpy as np
    from elprep_amd.engine import BqsrTables
    h = engine.header
    b = _priming_batch(h)
    engine.stage(b.take(np.arange(0, b.n // 2)))
    engine.stage(b.take(np.arange(b.n // 2, b.n)))
    engine.mark_duplicates(True)
    engine.sort_coordinate()
    engine.dup_metrics(100)
    if h.n_ref and h.n_rg and int(h.ref_len.sum()) <= 40_000_000:
        rng = np.random.default_rng(7)
        for r in range(h.n_ref):
            engine.set_reference(r, rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), int(h.ref_len[r])))
            engine.set_known_sites(r, np.zeros((0, 2), np.int32))
        qt, ct, xt = engine.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        engine.apply_bqsr(lut, present, 500)
    engine.sync()
    _STATE["primed"] += 1


def _install_reuse():
    from elprep_amd import engine as eng_mod
    E = eng_mod.Engine
    if getattr(E, "_reuse_installed", False):
        return
    real_init, real_close = E.__init__, E.close

    def init(self, header, device=0, flat_abi=False, tuning=None):
        if not _STATE["reuse"]:
            return real_init(self, header, device, flat_abi, tuning)
        key = _header_key(header, device, flat_abi)
        free = _POOL.setdefault(key, [])
        if free:
            donor = free.pop()
            self.__dict__.update(donor.__dict__)  # adopt the live context (handle, pinned buffers, header)
            donor.__dict__.clear()
            self.header = header
            self.reset()
            for k, v in _TUNE_DEFAULTS.items():
                self.set_tuning(k, v)
            for kv in filter(None, os.environ.get("ELP_TUNE", "").split(",")):
                k, v = kv.split("=")
                self.set_tuning(k.strip(), int(v))
            for k, v in (tuning or {}).items():
                self.set_tuning(k, v)
            _STATE["reused"] += 1
        else:
            real_init(self, header, device, flat_abi, None)
            _prime(self)
            self.reset()
            for kv in filter(None, os.environ.get("ELP_TUNE", "").split(",")):
                k, v = kv.split("=")
                self.set_tuning(k.strip(), int(v))
            for k, v in (tuning or {}).items():
                self.set_tuning(k, v)
        self._pool_key = key
        _OUT.append(self)

    def close(self):
        key = self.__dict__.get("_pool_key")
        if key is None or not getattr(self, "h", None) or not self.h.value:
            return real_close(self)
        if self in _OUT:
            _OUT.remove(self)
        keep = eng_mod.Engine.__new__(eng_mod.Engine)
        keep.__dict__.update(self.__dict__)
        self.__dict__.clear()
        _POOL.setdefault(key, []).append(keep)

    E.__init__, E.close, E._reuse_installed = init, close, True
    E._real_close = real_close


# tests that stay [fresh] only: several ranks / contexts wired to each other by callbacks that die with the test (group, transport,
# exchange, sfm), the compiled C++ consumer (its own process), and the two full-size property runs (minutes each)
_NO_REUSE = ("group_of_two", "transport", "exchange", "split_phase", "merge_phase", "sfm", "harness", "full_size", "c3_scale", "hg38", "allreduce")


def pytest_generate_tests(metafunc):
    if any(k in metafunc.definition.name for k in _NO_REUSE):
        return
    if metafunc.definition.get_closest_marker("gpu") and not metafunc.definition.get_closest_marker("fresh_only"):
        if "engine_mode" not in metafunc.fixturenames:
            metafunc.fixturenames.append("engine_mode")
        metafunc.parametrize("engine_mode", ["fresh", "reused"], indirect=True)


@pytest.fixture
def engine_mode(request):
    mode = getattr(request, "param", "fresh")
    if mode != "reused":
        yield mode
        return
    _install_reuse()
    _STATE["reuse"] = True
    try:
        yield mode
    finally:
        _STATE["reuse"] = False
        for e in list(_OUT):  # engines the test did not close (a failed assertion in front of close()) go back to the pool as well
            try:
                e.close()
            except Exception:
                pass
        del _OUT[:]


@pytest.fixture(scope="session", autouse=True)
def _drain_engine_pool():
    yield
    for free in _POOL.values():
        for e in free:
            try:
                e._real_close()
            except Exception:
                pass
    _POOL.clear()
    if _STATE["primed"] or _STATE["reused"]:
        sys.stderr.write("\n[reuse] pooled contexts primed: %d, hand-outs of a used context: %d\n" % (_STATE["primed"], _STATE["reused"]))
