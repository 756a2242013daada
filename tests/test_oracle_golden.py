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
