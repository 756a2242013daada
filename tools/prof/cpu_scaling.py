"""How the all-cores CPU baseline (oracle *_mt) scales on this box: threads -> Mreads/s per stage.  usage: cpu_scaling.py [reads]"""
import faulthandler, functools, os, sys, time
faulthandler.enable()
print = functools.partial(print, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as orc, bench
from tools import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "-")
cfg = synth.config("c3"); h = cfg.header()
from elprep_amd.batch import Batch
b = Batch.concat([synth.generate(cfg, lo, min(lo + 500_000, n // 2)) for lo in range(0, n // 2, 500_000)])
print("generated", b.n)
refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
sites = [bench.flatten_sites(synth.known_sites_raw(cfg, r)) for r in range(h.n_ref)]
ref = orc.BqsrRef(refs, sites)
for c in (4, 8, 16, 32, 64):
    if c > (os.cpu_count() or 1):
        break
    t = [time.perf_counter()]
    flags, _ = orc.dup_metrics_mt(b, h, None, 100, c); t.append(time.perf_counter())
    perm = orc.sort_coordinate_mt(b, flags, c); t.append(time.perf_counter())
    q = orc.bqsr_gather_mt(b, h, ref, flags, 500, c); t.append(time.perf_counter())
    fin = orc.BqsrFinal(*q, 500); t.append(time.perf_counter())
    orc.bqsr_apply_mt(fin, b, h, 0, (), c); t.append(time.perf_counter())
    d = [t[i + 1] - t[i] for i in range(5)]
    print(f"threads {c:4d}: markdup {d[0]:.2f}s sort {d[1]:.2f}s gather {d[2]:.2f}s finalize {d[3]:.2f}s apply {d[4]:.2f}s  total {sum(d):.2f}s = {b.n / sum(d) / 1e6:.2f} Mreads/s")
