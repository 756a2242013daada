"""CPU tests: the oracle against the hand-derived known-answer cases of tests/kat_cases.py (sr-tagged copies in the duplication
metrics; DeleteOrStore toggling for more than two primary records per QNAME) and the optical-duplicate list cap."""
import numpy as np

import oracle as orc
from tests import kat_cases


def test_sr_tagged_copies_are_dropped_before_the_metrics_pass():
    whole, splits, expected, dups = kat_cases.sr_case()
    h = kat_cases.header2()
    # `elprep filter` on everything
    perm = orc.sort_coordinate(whole)
    flags, ctr, _ = orc.dup_metrics(whole, h, perm, 100)
    assert ctr[0].tolist() == expected and ctr[1].sum() == 0
    assert kat_cases.flagged_names(whole, flags) == dups["filter"]
    # `elprep sfm`: one filter run per split file, counters summed (LoadAndCombineDuplicateMetrics, mark-optical-duplicates.go:711-731)
    tot = np.zeros_like(ctr)
    for name, sb in splits.items():
        perm = orc.sort_coordinate(sb)
        n_out = orc.num_sorted(sb)
        assert n_out == sb.n - int(sb.has_sr.sum()) and not sb.has_sr[perm[:n_out]].any()
        fl, c, _ = orc.dup_metrics(sb, h, perm, 100)
        assert kat_cases.flagged_names(sb, fl) == dups[name], name
        tot += c
    assert tot[0].tolist() == expected
    # hand-derived per split: group A counts its own pair once, the fragment and its unmapped mate; not the two tagged copies
    _, ca, _ = orc.dup_metrics(splits["A"], h, None, 100)
    assert ca[0].tolist() == [1, 1, 0, 1, 1, 0, 0]
    _, cs, _ = orc.dup_metrics(splits["spread"], h, None, 100)
    assert cs[0].tolist() == [0, 2, 0, 0, 0, 1, 0]


def test_delete_or_store_toggling_with_three_and_four_records_per_qname():
    h = kat_cases.header2()
    for k, (b, want) in enumerate(kat_cases.toggling_cases()):
        flags = orc.mark_duplicates(b, h)
        assert np.nonzero(flags & 0x400)[0].tolist() == want, k


def test_sort_keeps_only_records_without_sr():
    whole, splits, _, _ = kat_cases.sr_case()
    sb = splits["A"]
    perm = orc.sort_coordinate(sb)
    n_out = orc.num_sorted(sb)
    assert n_out == 4 and sorted(perm.tolist()) == list(range(sb.n))
    # p1 (100), p1 (300), then f1 and its unmapped mate at 500: the mapped read first (forward before ... both forward: QNAME ties,
    # flags 73 < 133), no tagged copy among them
    assert [sb.qname_of(i).decode() for i in perm[:n_out]] == ["p1", "p1", "f1", "f1"]
    assert [int(sb.flag[i]) for i in perm[:n_out]] == [99, 147, 73, 133]


def test_sort_runs_behind_mark_duplicates():
    b, h, order, dup = kat_cases.sort_sees_duplicate_bits_case()
    flags = orc.mark_duplicates(b, h)
    assert np.nonzero(flags & 0x400)[0].tolist() == dup
    assert orc.sort_coordinate(b, flags).tolist() == order
    assert orc.sort_coordinate(b).tolist() == [4, 0, 2, 5, 1, 3]  # without the duplicate bits the tie keeps staging order
