import sys, os, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np
from elprep_amd.engine import Engine
from tools import synth
reads = int(sys.argv[1])
cfg = synth.config("c3"); h = cfg.header()
engs = []
for k in range(2):
    e = Engine(h)
    for lo in range(0, reads // 2, 2_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
    e.snapshot(); engs.append(e)
A, B = engs
def t(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
def reset():
    for e in engs: e.rollback(); e.sync()
def sort(e): e.sort_coordinate(fetch=False); e.sync()
def md(e): e.mark_duplicates(True, fetch=False); e.sync()
for it in range(3):
    reset(); ts = t(lambda: sort(A)); tm = t(lambda: md(B))
    reset()
    def both():
        th = threading.Thread(target=sort, args=(A,)); th.start(); md(B); th.join()
    tb = t(both)
    reset()
    def both2():  # markdup + markdup
        th = threading.Thread(target=md, args=(A,)); th.start(); md(B); th.join()
    tmm = t(both2)
    print(f"iter {it}: sort {ts:.2f} ms, markdup {tm:.2f} ms, sequential {ts+tm:.2f}, concurrent {tb:.2f}; two markdups concurrent {tmm:.2f}", flush=True)
