"""Small driver for rocprofv3 sessions: stage N reads once, run the hot path `steps` times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from elprep_amd.engine import BqsrTables, Engine
from tools import synth
from bench import flatten_sites

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = synth.config("c3")
h = cfg.header()
e = Engine(h)
for lo in range(0, reads // 2, 2_000_000):
    e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
for r in range(h.n_ref):
    e.set_reference(r, synth.reference(cfg, r))
    e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
e.snapshot()
for s in range(steps):
    e.rollback()
    e.mark_duplicates(True, fetch=False)
    e.sort_coordinate(fetch=False)
    e.dup_metrics(100)
    qt, ct, xt = e.recalibrate(500)
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    e.apply_bqsr(lut, present, 500, fetch=False)
e.sync()
print("done", e.n)
