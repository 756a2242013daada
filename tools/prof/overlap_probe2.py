"""What a second compute stream could hide, measured without touching the library: two contexts on one GPU hold the
same reads; context A runs BQSR gather -> finalize -> apply, context B the coordinate sort and the duplication metrics - sequentially,
then from two host threads at the same time (every context has its own stream; ctypes releases the GIL inside a call).
usage: overlap_probe2.py <reads>"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import MAX_CYCLE, flatten_sites  # noqa: E402
from elprep_amd.engine import BqsrTables, Engine  # noqa: E402
from tools import synth  # noqa: E402

reads = int(sys.argv[1])
cfg = synth.config("c3")
h = cfg.header()
engs = []
for k in range(2):
    e = Engine(h)
    for lo in range(0, reads // 2, 2_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
    for r in range(h.n_ref):
        e.set_reference(r, synth.reference(cfg, r))
        e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
    e.snapshot()
    engs.append(e)
A, B = engs


def t(f):
    t0 = time.perf_counter()
    f()
    return (time.perf_counter() - t0) * 1e3


def reset():
    for e in engs:
        e.rollback()
        e.mark_duplicates(True, fetch=False)
        e.sync()


def bqsr(e):
    e.recalibrate_device(MAX_CYCLE)
    qt, ct, xt = e.tables_fetch(reuse=True)
    lut, present = BqsrTables(qt, ct, xt, MAX_CYCLE).finalize().build_lut(0)
    e.apply_bqsr(lut, present, MAX_CYCLE, fetch=False)
    e.sync()


def sort_metrics(e):
    e.sort_coordinate(fetch=False)
    e.dup_metrics(100)
    e.sync()


for it in range(4):
    reset()
    ta = t(lambda: bqsr(A))
    tb = t(lambda: sort_metrics(B))
    reset()

    def both():
        th = threading.Thread(target=sort_metrics, args=(B,))
        th.start()
        bqsr(A)
        th.join()
    tc = t(both)
    print(f"iter {it}: gather+finalize+apply {ta:.2f} ms, sort+metrics {tb:.2f} ms, one after the other {ta + tb:.2f} ms, at the same time {tc:.2f} ms "
          f"({ta + tb - tc:+.2f} ms hidden)", flush=True)
