"""The oracle against its own committed digests (tests/golden/oracle_regression.json, made by tests/golden/make_oracle_regression.py): the
checker every parity claim rests on must not change its answers unnoticed."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_regression as gold  # noqa: E402

_WANT = json.load(open(os.path.join(HERE, "golden", "oracle_regression.json")))["cases"]


@pytest.mark.parametrize("want", _WANT, ids=[f"seed{c['seed']}" for c in _WANT])
def test_oracle_outputs_match_the_committed_digests(want):
    got = gold.case(want["seed"], want["pairs"], want["qual_mode"], want["p_dup"], want["p_frag"])
    assert got["input"] == want["input"], "the synthetic generator changed: regenerate the fixture on purpose"
    for k in ("records", "duplicates", "observations", "mismatches", "upos", "score", "flags", "perm", "counters", "qual_table", "cycle_table", "context_table",
              "qual_out"):
        assert got[k] == want[k], k


# ---- reference-produced fixtures (tools/ref/make_fixtures.sh: the REAL elprep on the same synthetic reads).  They exist only where a Go
# toolchain was available to build elPrep; without them the oracle's parity stays "unpinned" and these tests are skipped.
_REF_DIR = os.path.join(HERE, "golden", "ref")
_REF_FIXTURES = sorted(f for f in (os.listdir(_REF_DIR) if os.path.isdir(_REF_DIR) else []) if f.endswith(".json"))


def _metrics_counters(text):
    """the per-library rows of the duplication metrics file: UNPAIRED_READS_EXAMINED READ_PAIRS_EXAMINED SECONDARY_OR_SUPPLEMENTARY_RDS
    UNMAPPED_READS UNPAIRED_READ_DUPLICATES READ_PAIR_DUPLICATES READ_PAIR_OPTICAL_DUPLICATES (filters/mark-optical-duplicates.go:608-626)"""
    rows, on = {}, False
    for line in text.splitlines():
        f = line.split("\t")
        if f and f[0] == "LIBRARY":
            on = True
            continue
        if on:
            if not line.strip():
                break
            rows[f[0]] = [int(x) for x in f[1:8]]
    return rows


@pytest.mark.skipif(not _REF_FIXTURES, reason="no reference-produced fixtures (tools/ref/make_fixtures.sh needs a Go toolchain): parity unpinned")
@pytest.mark.parametrize("name", _REF_FIXTURES)
def test_oracle_against_the_real_elprep(name):
    import numpy as np
    import hashlib
    import oracle as orc
    from tools import synth
    fix = json.load(open(os.path.join(_REF_DIR, name)))
    cfg = synth.config(fix["genome"], fix["seed_index"])
    cfg.p_frag = fix["p_frag"]
    b = synth.generate(cfg, 0, fix["pairs"])
    h = cfg.header()
    assert b.n == fix["records"]
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    flags = orc.mark_duplicates(b, h)
    perm = orc.sort_coordinate(b, flags)
    n_out = orc.num_sorted(b)
    assert np.array_equal(perm[:n_out], np.asarray(fix["order"], dtype=perm.dtype)), "coordinate order differs from elprep's output"
    assert np.array_equal(flags, np.asarray(fix["flags"], dtype=flags.dtype)), "FLAGs differ from elprep's output"
    _, ctr, _ = orc.dup_metrics(b, h, perm, 100)
    rows = _metrics_counters(fix["metrics_txt"])
    for lib, name_ in enumerate(h.lib_names):
        if name_ in rows:
            assert ctr[lib].tolist() == rows[name_], ("duplication metrics", name_)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    fin = orc.BqsrFinal(qt, ct, xt, 500)
    assert fin.report(h.cov_names, "GATK").strip() == fix["recal_txt"].strip(), "recalibration table differs from elprep's"
    qual = fin.apply(b, h, 0)
    lines = ["".join(chr(int(x) + 33) for x in qual[int(b.qual_off[i]):int(b.qual_off[i + 1])]) or "*" for i in range(b.n)]
    assert lines[:8] == fix["qual_head"]
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == fix["qual_sha256"], "recalibrated qualities differ from elprep's output"


def test_fixture_recipe_dry_run(tmp_path, monkeypatch):
    """tools/ref/make_fixtures.sh with the oracle standing in for the elprep binary: write_inputs.py -> the stand-in
    (out.sam, metrics.txt, recal.txt in elprep's formats) -> collect.py -> the comparison above, all in a temporary directory.  It proves
    the PLUMBING - that the SAM writer, the identification of the output's records and the parsing of the two text files work on the
    day somebody with a Go toolchain runs the real recipe; it pins nothing (the oracle is compared with itself) and leaves no fixture."""
    import subprocess
    root = os.path.dirname(HERE)
    work, fixdir = tmp_path / "w", tmp_path / "ref"
    fixdir.mkdir()
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "ref", "write_inputs.py"), str(work), "1500", "0"])
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "ref", "oracle_as_elprep.py"), str(work)])
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "ref", "collect.py"), str(work), str(fixdir / "filter_tiny_seed0.json")])
    fix = json.load(open(fixdir / "filter_tiny_seed0.json"))
    assert fix["records"] > 2900 and len(fix["order"]) <= fix["records"] and sum(1 for x in fix["flags"] if x & 0x400) > 50
    assert "LIBRARY" in fix["metrics_txt"] and "#:GATKTable" in fix["recal_txt"]
    monkeypatch.setattr(sys.modules[__name__], "_REF_DIR", str(fixdir))
    test_oracle_against_the_real_elprep.__wrapped__("filter_tiny_seed0.json") if hasattr(test_oracle_against_the_real_elprep, "__wrapped__") else \
        test_oracle_against_the_real_elprep("filter_tiny_seed0.json")
    assert not os.path.isdir(os.path.join(HERE, "golden", "ref")) or "filter_tiny_seed0.json" not in os.listdir(os.path.join(HERE, "golden", "ref"))
