"""elp_stage_bgzf / elp_emit_sorted_bgzf on N reads: wall time and per-kernel times.  usage: bgzf_speed.py [reads] [zlib level] [check]
(check: the staged records give the flags, the permutation and the sorted BAM bytes that elp_stage_bam gives on the inflated bytes)"""
import struct
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elprep_amd.engine import Engine  # noqa: E402
from tools import synth  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = synth.config("c3")
h = cfg.header()
b = synth.generate(cfg, 0, reads // 2)
raw, rec_off = synth.bam_records(b, h.rg_ids)
raw = raw.tobytes()
cut = 65280


def member(k):
    part = raw[k:k + cut]
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    data = co.compress(part) + co.flush()
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(data) + 25) + data + struct.pack("<II", zlib.crc32(part), len(part))


with ThreadPoolExecutor(16) as pool:
    bz = np.frombuffer(b"".join(pool.map(member, range(0, len(raw), cut))), dtype=np.uint8)
e = Engine(h, 0)
e.set_read_group_ids(h.rg_ids)
try:
    e.stage_bgzf(bz)
except Exception as ex:
    print("stage_bgzf:", str(ex)[:80])
e.sync()
for _ in range(2):
    e.reset()
    e.profile_enable(True)
    e.profile_reset()
    t0 = time.perf_counter()
    try:
        e.stage_bgzf(bz)
    except Exception as ex:
        print("stage_bgzf:", str(ex)[:80])
    e.sync()
    t = time.perf_counter() - t0
    prof = e.profile()
    e.profile_enable(False)
print(f"{b.n} reads, {len(raw)} inflated bytes, {bz.size} compressed: stage_bgzf {t * 1e3:.1f} ms = {b.n / t / 1e6:.1f} Mreads/s = {len(raw) / t / 1e9:.2f} GB/s inflated")
print({k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:10]})
if len(sys.argv) > 3 and sys.argv[3] == "check":
    import hashlib
    got = (e.n, e.mark_duplicates(True), e.sort_coordinate(), hashlib.sha1(e.emit_sorted_bam().tobytes()).hexdigest())
    e2 = Engine(h, 0)
    e2.set_read_group_ids(h.rg_ids)
    e2.stage_bam(np.frombuffer(raw, dtype=np.uint8), rec_off=rec_off)
    want = (e2.n, e2.mark_duplicates(True), e2.sort_coordinate(), hashlib.sha1(e2.emit_sorted_bam().tobytes()).hexdigest())
    e2.close()
    ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[3] == want[3]
    print("check against elp_stage_bam:", "identical" if ok else "DIFFERENT", got[0], want[0])
else:
    e.mark_duplicates(True, fetch=False)
    e.sort_coordinate(fetch=False)
e.emit_sorted_bgzf()
e.profile_enable(True)
e.profile_reset()
t0 = time.perf_counter()
out = e.emit_sorted_bgzf()
t = time.perf_counter() - t0
prof = e.profile()
e.profile_enable(False)
print(f"emit_sorted_bgzf {t * 1e3:.1f} ms = {b.n / t / 1e6:.1f} Mreads/s ({out.size} bytes = {out.size / len(raw):.3f} of the inflated size)")
print({k: round(v[1], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]})
