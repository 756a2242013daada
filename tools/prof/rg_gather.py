"""BQSR gather (+ apply record passes) kernel times with many read groups: usage rg_gather.py [reads] [n_lanes ...]
(ELP_HIP_SO selects the library: timing probes of the prologue's append were run through this)"""
import sys

sys.path.insert(0, ".")
from elprep_amd.engine import Engine  # noqa: E402
from tools import synth  # noqa: E402
from bench import flatten_sites  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
lanes = [int(a) for a in sys.argv[2:]] or [4, 16, 32]
for nl in lanes:
    cfg = synth.config("c3")
    cfg.n_lanes = nl
    h = cfg.header()
    e = Engine(h)
    for lo in range(0, reads // 2, 1_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 1_000_000, reads // 2)))
    for r in range(h.n_ref):
        e.set_reference(r, synth.reference(cfg, r))
        e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
    e.sync()
    e.snapshot()
    e.mark_duplicates(True, fetch=False)
    best = None
    for it in range(4):
        e.profile_enable(True)
        e.profile_reset()
        e.recalibrate_device(150)
        e.sync()
        prof = e.profile()
        e.profile_enable(False)
        tot = sum(v[1] for v in prof.values())
        if best is None or tot < best[0]:
            best = (tot, prof)
    print(f"{nl} read groups, {e.n} reads: gather kernels {best[0]:.3f} ms", {k: round(v[1], 3) for k, v in sorted(best[1].items(), key=lambda kv: -kv[1][1])[:10]})
    e.close()
