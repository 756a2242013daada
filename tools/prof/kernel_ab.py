"""A/B timing of the BQSR kernels across several builds of libelprep_hip.so (one subprocess per library).
usage: kernel_ab.py <reads> <lib.so> [<lib.so> ...]      prints per library the per-kernel ms of gather + apply (best of 3)"""
import os
import subprocess
import sys

if os.environ.get("ELP_AB_LIB"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from elprep_amd import _lib
    _lib.HIP_SO = os.environ["ELP_AB_LIB"]
    from elprep_amd.engine import BqsrTables, Engine
    from tools import synth
    from bench import flatten_sites
    reads = int(sys.argv[1])
    cfg = synth.config("c3")
    h = cfg.header()
    e = Engine(h)
    for lo in range(0, reads // 2, 2_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
    for r in range(h.n_ref):
        e.set_reference(r, synth.reference(cfg, r))
        e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
    e.snapshot()
    best = {}
    for it in range(4):
        e.rollback(); e.sync()
        e.profile_enable(True); e.profile_reset()
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        e.apply_bqsr(lut, present, 500, fetch=False)
        e.sync()
        prof = e.profile(); e.profile_enable(False)
        if it == 0:
            continue
        for k, (c, ms) in prof.items():
            best[k] = min(best.get(k, 1e9), ms)
    print(os.path.basename(os.environ["ELP_AB_LIB"]), " ".join(f"{k}={v:.3f}" for k, v in sorted(best.items(), key=lambda kv: -kv[1])[:5]), flush=True)
else:
    for lib in sys.argv[2:]:
        env = dict(os.environ, ELP_AB_LIB=os.path.abspath(lib))
        subprocess.call([sys.executable, os.path.abspath(__file__), sys.argv[1]], env=env)
