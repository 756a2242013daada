"""The shuffled-order workload of bench.py (reads in random order within every 2 M-record batch), the path in SERIAL order with the
context's profile: per-kernel times without the three chains sharing the chip.  usage: shuffled_serial.py [reads] [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from elprep_amd.engine import BqsrTables, Engine
from tools import synth
from bench import flatten_sites

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = synth.config("c3")
h = cfg.header()
e = Engine(h)
rng = np.random.default_rng(7)
for lo in range(0, reads // 2, 1_000_000):
    b = synth.generate(cfg, lo, min(lo + 1_000_000, reads // 2))
    e.stage(b.take(rng.permutation(b.n)))
for r in range(h.n_ref):
    e.set_reference(r, synth.reference(cfg, r))
    e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
e.snapshot()
for s in range(steps + 1):
    e.rollback()
    if s == 1:
        e.profile_enable(True)
        e.profile_reset()
        e.sync()
        t0 = time.perf_counter()
    e.mark_duplicates(True, fetch=False)
    e.sort_coordinate(fetch=False)
    e.dup_metrics(100)
    qt, ct, xt = e.recalibrate(500)
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    e.apply_bqsr(lut, present, 500, fetch=False)
e.sync()
t = (time.perf_counter() - t0) / steps
prof = e.profile()
print(f"{e.n} reads shuffled, serial order: {t * 1e3:.2f} ms per step (rollback included)")
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    if v[1] / steps >= 0.03:
        print(f"  {k:28s} {v[0] / steps:6.1f} launches  {v[1] / steps:7.3f} ms")
    tot += v[1] / steps
print(f"  kernels total {tot:.2f} ms")
