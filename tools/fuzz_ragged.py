"""More seeds of the adversarial BQSR cases of tests/test_gpu_ragged.py (ragged reads, clips, up to four indels, adaptor geometry, low
quality tails, dense known sites) - run on the GPU box.  usage: python tools/fuzz_ragged.py [first_seed] [n_seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_ragged import _check_gather_apply, _random_case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    quals = [2, 5, 6, 12, 23, 37, 41] if seed % 3 else list(range(2, 45))
    n_cov = 2 + seed % 3
    b, h, refs, sites = _random_case(seed, 3000 + 500 * (seed % 7), quals=quals, n_cov=n_cov)
    try:
        _check_gather_apply(b, h, refs, sites, chunks=1 + seed % 3)
        print(f"seed {seed}: {b.n} records, {len(quals)} qualities, {n_cov} covariates: ok", flush=True)
    except AssertionError as ex:
        bad += 1
        print(f"seed {seed}: MISMATCH {ex}", flush=True)
print("mismatching seeds:", bad)
sys.exit(1 if bad else 0)
