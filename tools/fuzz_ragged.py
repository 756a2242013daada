"""More seeds of the adversarial BQSR cases of tests/test_gpu_ragged.py (ragged reads, clips, up to four indels, adaptor geometry, low
quality tails, dense known sites) - run on the GPU box.  usage: python tools/fuzz_ragged.py [first_seed] [n_seeds] [one]
"one": every read of a seed has ONE length (17 .. 260, drawn per seed) - the dispatch takes count3 / apply3 (round 3)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ragged import _check_gather_apply, _random_case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
one = len(sys.argv) > 3 and sys.argv[3] == "one"
bad = 0
for seed in range(first, first + count):
    quals = [2, 5, 6, 12, 23, 37, 41] if seed % 3 else list(range(2, 45))
    n_cov = (2 + seed % 3) if seed % 4 else [17, 33, 70, 24][(seed // 4) % 4]  # (round 5: every fourth seed many read groups)
    if one:
        length = int(np.random.default_rng(seed).integers(17, 261))
        b, h, refs, sites = _random_case(seed, 3000 + 500 * (seed % 7), quals=quals, n_cov=n_cov, len_mix=((length, length, 1.0),))
        b = b.take(np.arange(b.n - 2))  # the generator's two odd records have other lengths
        assert len(set(np.diff(b.qual_off).tolist())) == 1
    else:
        b, h, refs, sites = _random_case(seed, 3000 + 500 * (seed % 7), quals=quals, n_cov=n_cov)
    try:
        _check_gather_apply(b, h, refs, sites, chunks=1 + seed % 3)
        print(f"seed {seed}: {b.n} records{f' of {length} bases' if one else ''}, {len(quals)} qualities, {n_cov} covariates: ok", flush=True)
    except AssertionError as ex:
        bad += 1
        print(f"seed {seed}: MISMATCH {ex}", flush=True)
print("mismatching seeds:", bad)
sys.exit(1 if bad else 0)
