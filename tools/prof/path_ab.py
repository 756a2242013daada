"""A/B timing of the whole hot path across several builds of libelprep_hip.so (one subprocess per library, same box, same input).
usage: path_ab.py <reads> <lib.so> [<lib.so> ...] [--quals full]
prints per library: best-of-3 per-kernel ms (top kernels), the per-stage sums and the checksum of (flags, permutation, new
qualities) so that the builds can be seen to agree."""
import os
import subprocess
import sys
import zlib

if os.environ.get("ELP_AB_LIB"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from elprep_amd import _lib
    _lib.HIP_SO = os.environ["ELP_AB_LIB"]
    import numpy as np
    from elprep_amd.engine import BqsrTables, Engine
    from tools import synth
    from bench import flatten_sites
    reads = int(sys.argv[1])
    cfg = synth.config("c3")
    if "--quals" in sys.argv and sys.argv[sys.argv.index("--quals") + 1] == "full":
        cfg.qual_mode = 1
    h = cfg.header()
    e = Engine(h)
    for lo in range(0, reads // 2, 2_000_000):
        e.stage(synth.generate(cfg, lo, min(lo + 2_000_000, reads // 2)))
    for r in range(h.n_ref):
        e.set_reference(r, synth.reference(cfg, r))
        e.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
    e.snapshot()
    best = {}
    crc = 0
    for it in range(4):
        e.rollback(); e.sync()
        e.profile_enable(True); e.profile_reset()
        e.mark_duplicates(True, fetch=False)
        e.sort_coordinate(fetch=False)
        ctr = e.dup_metrics(100)
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        e.apply_bqsr(lut, present, 500, fetch=False)
        e.sync()
        prof = e.profile(); e.profile_enable(False)
        if it == 0:
            crc = zlib.crc32(np.ascontiguousarray(e.flags()).tobytes())
            crc = zlib.crc32(np.ascontiguousarray(e.permutation()).tobytes(), crc)
            crc = zlib.crc32(np.ascontiguousarray(e.qual()).tobytes(), crc)
            crc = zlib.crc32(np.ascontiguousarray(ctr).tobytes(), crc)
            crc = zlib.crc32(np.ascontiguousarray(ct).tobytes(), crc)
            continue
        for k, (c, ms) in prof.items():
            best[k] = min(best.get(k, 1e9), ms)
    tot = sum(best.values())
    print(os.path.basename(os.environ["ELP_AB_LIB"]), f"crc={crc:08x} kernels_total={tot:.3f}",
          " ".join(f"{k}={v:.3f}" for k, v in sorted(best.items(), key=lambda kv: -kv[1]) if v >= 0.03), flush=True)
else:
    args = sys.argv[1:]
    extra = []
    if "--quals" in args:
        i = args.index("--quals"); extra = args[i:i + 2]; del args[i:i + 2]
    for lib in args[1:]:
        env = dict(os.environ, ELP_AB_LIB=os.path.abspath(lib))
        subprocess.call([sys.executable, os.path.abspath(__file__), args[0]] + extra, env=env)
