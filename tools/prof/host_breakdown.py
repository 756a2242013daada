"""Wall-clock per host call of one step (includes the device work each call waits for): where the non-kernel time goes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from elprep_amd.engine import BqsrTables, Engine
from tools import synth
from bench import flatten_sites

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
cfg = synth.config("c3")
h = cfg.header()
e = Engine(h)
for lo in range(0, reads // 2, 1_000_000):
    e.stage(synth.generate(cfg, lo, min(lo + 1_000_000, reads // 2)))
for r in range(h.n_ref):
    e.set_reference(r, synth.reference(cfg, r))
    e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
e.snapshot()
for it in range(3):
    e.rollback(); e.sync()
    e.profile_enable(True); e.profile_reset()
    t = [time.perf_counter()]
    def lap(): e.sync(); t.append(time.perf_counter())
    e.sort_coordinate(fetch=False); lap()
    e.mark_duplicates(True, fetch=False); lap()
    e.dup_metrics(100); lap()
    qt, ct, xt = e.recalibrate(500, reuse=True); lap()
    tb = BqsrTables(qt, ct, xt, 500); lap()
    tb.finalize(); lap()
    lut, present = tb.build_lut(0); lap()
    e.apply_bqsr(lut, present, 500, fetch=False); lap()
    prof = e.profile(); e.profile_enable(False)
    names = ["sort", "markdup", "metrics", "gather", "tables_new", "finalize", "build_lut", "apply"]
    d = np.diff(t) * 1e3
    kms = {}
    for k, (c, ms) in prof.items():
        st = "sort" if k.startswith(("radix", "scan", "tie", "iota", "material", "seg", "large", "add_own", "adapt", "flat")) else "markdup" if k.startswith("md_") else "metrics" if k.startswith("mx_") else "apply" if k.startswith("bqsr_apply") else "gather"
        kms[st] = kms.get(st, 0) + ms
    print("iter", it, " ".join(f"{n}={x:.2f}ms(k={kms.get(n,0):.2f})" for n, x in zip(names, d)), "total=%.2f" % d.sum())
