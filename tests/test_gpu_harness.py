"""The compiled C++ consumer of the two C ABIs (tests/host_harness.cpp): built on CPU (it must compile and link against both shared
objects), run on the GPU box against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as orc
from elprep_amd import _lib
from tests.common import dataset

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host_harness")


def build_harness():
    src = os.path.join(HERE, "host_harness.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(src) > os.path.getmtime(EXE):
        pkg = os.path.dirname(_lib.HIP_SO)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", EXE, src, "-L" + pkg, "-lelprep_hip", "-lelprep_host", "-lpthread",
                               "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return EXE


def test_harness_compiles_and_links():
    assert os.path.exists(build_harness())


def _pad8(a: np.ndarray) -> bytes:
    raw = np.ascontiguousarray(a).tobytes()
    return raw + b"\0" * ((8 - len(raw) % 8) % 8)


@pytest.mark.gpu
def test_harness_runs_the_filter_sequence(tmp_path):
    cfg, b, h, refs, sites = dataset("tiny", 5000, 2, 0.03)
    parts = [np.asarray([h.n_ref, h.n_rg, h.n_lib, h.n_cov, 500, 100], np.int32).tobytes(), np.asarray([b.n], np.uint64).tobytes(),
             _pad8(h.ref_len), _pad8(h.rg_lib), _pad8(h.rg_cov)]
    for name in ("refid", "pos", "next_refid", "pnext", "tlen", "flag", "mapq", "rgid", "has_sr", "l_seq", "qname_off", "qname", "cigar_off", "cigar",
                 "seq_off", "seq4", "qual_off", "qual"):
        parts.append(_pad8(getattr(b, name)))
    for r in range(h.n_ref):
        parts += [np.asarray([refs[r].size], np.int64).tobytes(), _pad8(refs[r]), np.asarray([sites[r].shape[0]], np.int64).tobytes(),
                  _pad8(np.ascontiguousarray(sites[r], dtype=np.int32))]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    open(fin, "wb").write(b"".join(parts))
    res = subprocess.run([build_harness(), fin, fout, "3"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(fout, dtype=np.uint8)
    n_sorted, qb, rl = (int(x) for x in raw[:24].view(np.uint64))
    at = 24
    def take(dt, cnt):
        nonlocal at
        a = raw[at:at + cnt * np.dtype(dt).itemsize].view(dt)
        at += cnt * np.dtype(dt).itemsize
        return a
    perm, flags = take(np.uint32, b.n), take(np.uint16, b.n)
    ctr = take(np.int64, (h.n_lib + 1) * 7).reshape(h.n_lib + 1, 7)
    qt = take(np.int64, h.n_cov * 94 * 2).reshape(h.n_cov, 94, 2)
    ct = take(np.int64, h.n_cov * 94 * 1001 * 2).reshape(h.n_cov, 94, 1001, 2)
    xt = take(np.int64, h.n_cov * 94 * 16 * 2).reshape(h.n_cov, 94, 16, 2)
    qual, report = take(np.uint8, qb), bytes(take(np.uint8, rl)).decode()
    oflags = orc.mark_duplicates(b, h)
    operm = orc.sort_coordinate(b, oflags)
    assert n_sorted == b.n and np.array_equal(flags, oflags) and np.array_equal(perm, operm)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    assert np.array_equal(ctr, octr)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    fin_ = orc.BqsrFinal(oq, oc, ox, 500)
    assert np.array_equal(qual, fin_.apply(b, h, 0))
    assert report == fin_.report([f"cov{k}" for k in range(h.n_cov)], "GATK")
