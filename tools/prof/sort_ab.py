"""A/B of the sort stage across builds of libelprep_hip.so: usage sort_ab.py <reads> <lib.so>..."""
import os, subprocess, sys, time
if os.environ.get("ELP_AB_LIB"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from elprep_amd import _lib
    _lib.HIP_SO = os.environ["ELP_AB_LIB"]
    from elprep_amd.engine import Engine
    from tools import synth
    reads = int(sys.argv[1])
    cfg = synth.config("c3"); h = cfg.header()
    e = Engine(h)
    for lo in range(0, reads // 2, 2_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
    e.snapshot()
    e.sort_coordinate(fetch=False); e.sync()
    best = 1e9
    for it in range(5):
        e.rollback(); e.sync()
        e.mark_duplicates(True, fetch=False); e.sync()   # adapt happens here, outside the timed call
        t0 = time.perf_counter(); e.sort_coordinate(fetch=False); e.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(os.path.basename(os.environ["ELP_AB_LIB"]), "sort_coordinate %.3f ms" % best, flush=True)
else:
    for lib in sys.argv[2:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), sys.argv[1]], env=dict(os.environ, ELP_AB_LIB=os.path.abspath(lib)))
