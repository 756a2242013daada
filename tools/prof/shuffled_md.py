"""mark duplicates on reads staged in random order (mates are not neighbours): the partitioned mate pass against the table in HBM.
usage: shuffled_md.py [reads]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from elprep_amd.engine import Engine  # noqa: E402
from tools import synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
cfg = synth.config("c3")
h = cfg.header()
rng = np.random.default_rng(7)
parts = []
for lo in range(0, reads // 2, 1_000_000):
    b = synth.generate(cfg, lo, min(lo + 1_000_000, reads // 2))
    parts.append(b.take(rng.permutation(b.n)))
for path in (0, 2, 1):
    e = Engine(h, tuning={"mate_path": path})
    for p in parts:
        e.stage(p)
    e.snapshot()
    best, prof = 1e9, None
    for it in range(3):
        e.rollback()
        e.sync()
        e.profile_enable(True)
        e.profile_reset()
        t0 = time.perf_counter()
        e.mark_duplicates(True, fetch=False)
        e.sync()
        t = time.perf_counter() - t0
        if t < best:
            best, prof = t, e.profile()
        e.profile_enable(False)
    fl = e.flags()
    print(f"mate_path {path}: {best * 1e3:.2f} ms, dups {int(((fl & 0x400) != 0).sum())}", {k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]})
    e.close()
