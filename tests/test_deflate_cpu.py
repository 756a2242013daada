"""CPU tests of the device's DEFLATE compressor through its host emulation (tests/deflate_host.cpp runs the functions of
elprep_amd/csrc/deflate_core.hpp that the kernel k_bgzf_deflate runs): zlib inflates every block to exactly its payload - BAM records of
the synthetic workload, runs, random bytes (stored fallback), short and empty tails - in both part orders."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "libdeflate_host.so")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(ROOT, "tests", "deflate_host.cpp")
    hdr = os.path.join(ROOT, "elprep_amd", "csrc", "deflate_core.hpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, src])
    L = C.CDLL(SO)
    L.dfl_emulate_block.restype = C.c_uint32
    L.dfl_emulate_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    return L


def _deflate(L, data: bytes, order: int):
    a = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
    out = np.zeros(len(data) + 64, np.uint8)
    kind, ntok = C.c_int(), C.c_uint32()
    n = L.dfl_emulate_block(a.ctypes.data, len(data), out.ctypes.data, order, C.byref(kind), C.byref(ntok))
    return out[:n].tobytes(), int(kind.value), int(ntok.value)  # kind: 0 fixed codes, 1 stored, 2 dynamic codes


def _inflate(cdata: bytes) -> bytes:
    d = zlib.decompressobj(-15)
    out = d.decompress(cdata) + d.flush()
    assert d.eof and not d.unused_data
    return out


def _cases():
    from tools import synth
    rng = np.random.default_rng(1)
    cfg = synth.config("c3")
    b = synth.generate(cfg, 0, 1500)
    bam, _ = synth.bam_records(b, cfg.header().rg_ids)
    bam = bam.tobytes()
    yield "bam block", bam[:65280]
    yield "bam block 2", bam[65280:2 * 65280]
    yield "short tail", bam[:30011]
    yield "one part and a bit", bam[:300]
    yield "three bytes", b"abc"
    yield "empty", b""
    yield "zeros", bytes(65280)
    yield "one run per part", bytes([k % 251 for k in range(256) for _ in range(255)])
    yield "random", rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
    yield "text", (b"SIM0:1:FC1:1:1656:18011:10371 " * 3000)[:65280]
    yield "high literals", bytes(rng.integers(144, 256, 5000, dtype=np.uint8)) + bytes(200) + bytes(rng.integers(250, 256, 4000, dtype=np.uint8))
    yield "far matches", (rng.integers(0, 256, 20000, dtype=np.uint8).tobytes() * 4)[:65280]


@pytest.mark.parametrize("order", [0, 1, 16, 17])  # (+ 16: fixed codes only - round 5's form, what a block falls back to)
def test_every_block_inflates_to_its_payload(lib, order):
    ratios = {}
    for name, data in _cases():
        cdata, kind, ntok = _deflate(lib, data, order)
        assert _inflate(cdata) == data, name
        assert len(cdata) <= len(data) + 5, name
        ratios[name] = (round(len(cdata) / max(len(data), 1), 3), kind, ntok)
    assert ratios["random"][1] == 1 and ratios["bam block"][1] == (2 if order < 16 else 0)
    assert ratios["zeros"][0] < 0.05
    # the synthetic BAM records: zlib's own fixed-Huffman level-1 encoder reaches 0.55 on them; the strip-parallel match finder reaches the same
    assert ratios["bam block"][0] < 0.58, ratios
    if order < 16:  # dynamic codes (round 6): zlib level 1 - dynamic codes over its own parse - reaches 0.448 on these records
        assert ratios["bam block"][0] < 0.45, ratios
        z = zlib.compressobj(1, zlib.DEFLATED, -15)
        data = dict(_cases())["bam block"]
        assert len(cdata) >= 0 and ratios["bam block"][0] * len(data) < len(z.compress(data) + z.flush())


def test_dynamic_codes_at_the_edges(lib):
    """alphabets of one and two symbols, no distance code at all, a count distribution that wants codes longer than 15 bits (Fibonacci
    counts: the length limit's repair), long runs of unused symbols (run-length symbols 17 and 18 of the header) - every block inflates"""
    rng = np.random.default_rng(5)
    fib = [1, 1]
    while sum(fib) < 40000:
        fib.append(fib[-1] + fib[-2])
    skew = b"".join(bytes([7 * k % 256]) * f for k, f in enumerate(fib))
    skew = bytes(rng.permutation(np.frombuffer(skew, dtype=np.uint8)))[:65280]
    cases = {
        "one literal": b"a",
        "two literals": b"ab",
        "literals only": bytes(rng.permutation(np.arange(200, dtype=np.uint8))) * 3,
        "one symbol many times": b"x" * 3 + b"y",
        "skewed counts": skew,
        "two values": bytes(rng.integers(0, 2, 65280, dtype=np.uint8) * 255),
        "sparse alphabet": bytes(rng.choice(np.array([0, 1, 2, 127, 128, 254, 255], dtype=np.uint8), 30000)),
    }
    kinds = {}
    for name, data in cases.items():
        for order in (0, 1):
            cdata, kind, _ = _deflate(lib, data, order)
            assert _inflate(cdata) == data, (name, order)
            kinds[name] = kind
    assert kinds["skewed counts"] == 2 and kinds["two values"] == 2 and kinds["sparse alphabet"] == 2, kinds


def test_symbol_tables_against_the_rfc(lib):
    """match lengths and distances through the encoder and zlib's decoder: random bytes, then a copy of `length` bytes from `dist` back
    that starts a strip of the match finder (its source is in the table by then) and lies inside one part of the parse"""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    seen = set()
    for dist in [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16, 17, 24, 25, 32, 33, 48, 49, 64, 65, 96, 97, 128, 129, 192, 193, 256, 257, 384, 385, 512, 513, 768, 769, 1024,
                 1025, 1536, 1537, 2048, 2049, 3072, 3073, 4096, 4097, 6144, 6145, 8192, 8193, 12288, 12289, 16384, 16385, 24576, 24577, 32767, 32768]:
        for length in (4, 5, 10, 11, 12, 13, 18, 19, 34, 35, 66, 67, 115, 130, 131, 226, 227, 250, 254):
            k = max((dist + 255) // 256, 1)       # the copy starts strip k, i.e. at offset k of part k
            if k + length > 255:
                continue
            pre = 256 * k
            data = bytearray(base[:pre])
            for j in range(length):
                data.append(data[pre - dist + j])
            data += base[60000:60050]
            for order in (0, 1):
                cdata, stored, _ = _deflate(lib, bytes(data), order)
                assert _inflate(cdata) == bytes(data), (dist, length, order)
            seen.add((dist, length))
    assert len(seen) > 500


def test_code_lengths_are_huffman_lengths(lib):
    """the builder's code lengths against a textbook Huffman construction (heap): the same total cost whenever the tree fits the limit,
    a complete code (Kraft sum exactly 1) always - also where the limit bites (Fibonacci counts, limit 7 as for the code of the code
    lengths: the first form of the repair left such a code over-subscribed by 3 / 32768 - found by this test) -, and near the unlimited
    optimum there"""
    import heapq
    lib.dfl_code_lengths.restype = None
    lib.dfl_code_lengths.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(9)

    def textbook(freq):
        heap = [(int(f), k, (k,)) for k, f in enumerate(freq) if f]
        depth = {k: 0 for _, k, _ in heap}
        heapq.heapify(heap)
        tie = len(freq)
        while len(heap) > 1:
            fa, _, sa = heapq.heappop(heap)
            fb, _, sb = heapq.heappop(heap)
            for k in sa + sb:
                depth[k] += 1
            heapq.heappush(heap, (fa + fb, tie, sa + sb))
            tie += 1
        return depth

    cases = []
    for n, maxbits in ((286, 15), (30, 15), (19, 7)):
        for _ in range(30):
            f = rng.integers(0, 2000, n).astype(np.uint32)
            f[rng.random(n) < rng.random() * 0.8] = 0
            if (f != 0).sum() < 2:
                f[:2] = 5
            cases.append((f, maxbits))
        fib = [1, 1]
        while len(fib) < min(n, 24):
            fib.append(fib[-1] + fib[-2])
        f = np.zeros(n, np.uint32)
        f[:len(fib)] = fib
        cases.append((rng.permutation(f).astype(np.uint32), maxbits))
    limited = 0
    for f, maxbits in cases:
        out = np.zeros(f.size, np.uint8)
        fa = np.ascontiguousarray(f)
        lib.dfl_code_lengths(fa.ctypes.data, int(f.size), maxbits, out.ctypes.data)
        used = f != 0
        assert (out[used] >= 1).all() and (out[~used] == 0).all() and out.max() <= maxbits
        assert sum(2.0 ** -int(l) for l in out[used]) == 1.0  # a complete code
        d = textbook(f)
        best = sum(int(f[k]) * v for k, v in d.items())
        cost = int((f.astype(np.int64) * out).sum())
        if max(d.values()) <= maxbits:
            assert cost == best, (cost, best)
        else:
            limited += 1
            assert best <= cost <= best * 1.3, (cost, best, maxbits)  # (Fibonacci counts on 19 symbols under a 7-bit limit: +12 %)
    assert limited >= 3


def test_block_header_reads_back(lib):
    """the run-length coded header (build_header / emit_dyn_header) parsed back by a reader written from RFC 1951 3.2.7: random length
    vectors with long runs of unused symbols (symbols 17 / 18, runs longer than 138), repeats (symbol 16, runs longer than 6), trailing
    zeros (HLIT / HDIST trimmed), a single distance code"""
    lib.dfl_header_bits.restype = C.c_uint32
    lib.dfl_header_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(21)

    def complete_lengths(n_used, n, maxbits):
        """a random complete prefix code on n_used of n symbols (split the Kraft budget at random)"""
        lens = [1, 1]
        while len(lens) < n_used:
            k = int(rng.integers(0, len(lens)))
            if lens[k] >= maxbits:
                if all(l >= maxbits for l in lens):
                    break
                continue
            lens[k] += 1
            lens.insert(k, lens[k])
        out = np.zeros(n, np.uint8)
        pos = np.sort(rng.choice(n, len(lens), replace=False)) if rng.random() < 0.5 else np.arange(len(lens)) + int(rng.integers(0, n - len(lens) + 1))
        out[pos] = rng.permutation(np.asarray(lens, np.uint8)) if rng.random() < 0.5 else np.sort(np.asarray(lens, np.uint8))
        return out

    def read_header(buf):
        bits = int.from_bytes(buf, "little")
        at = [0]

        def take(k):
            v = (bits >> at[0]) & ((1 << k) - 1)
            at[0] += k
            return v
        assert take(1) == 1 and take(2) == 2
        hlit, hdist, hclen = take(5) + 257, take(5) + 1, take(4) + 4
        order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
        cl = [0] * 19
        for k in range(hclen):
            cl[order[k]] = take(3)
        # canonical decoding table of the code-length code
        code, table = 0, {}
        for length in range(1, 8):
            for sym in range(19):
                if cl[sym] == length:
                    table[(length, code)] = sym
                    code += 1
            code <<= 1
        used = [l for l in cl if l]
        assert sum(2.0 ** -l for l in used) == 1.0 or len(used) == 1 or used == [1, 1]

        def symbol():
            c = 0
            for length in range(1, 8):
                c = (c << 1) | take(1)
                if (length, c) in table:
                    return table[(length, c)]
            raise AssertionError("no code")
        lens = []
        while len(lens) < hlit + hdist:
            s = symbol()
            if s < 16:
                lens.append(s)
            elif s == 16:
                assert lens
                lens += [lens[-1]] * (3 + take(2))
            elif s == 17:
                lens += [0] * (3 + take(3))
            else:
                lens += [0] * (11 + take(7))
        assert len(lens) == hlit + hdist
        return lens[:hlit], lens[hlit:], at[0]

    for trial in range(120):
        ll = complete_lengths(int(rng.integers(2, 287)), 286, 15)
        if ll[256] == 0:  # the end-of-block symbol is always used
            j = int(np.nonzero(ll)[0][0])
            ll[256], ll[j] = ll[j], 0
        nd = int(rng.integers(0, 31))
        if nd < 2:
            d = np.zeros(30, np.uint8)
            d[:2] = 1
        else:
            d = complete_lengths(nd, 30, 15)
        out = np.zeros(1024, np.uint8)
        nbits = lib.dfl_header_bits(np.ascontiguousarray(ll).ctypes.data, np.ascontiguousarray(d).ctypes.data, out.ctypes.data, 1024)
        got_ll, got_d, read = read_header(out.tobytes())
        assert read == nbits
        assert got_ll == [int(x) for x in ll[:len(got_ll)]] and not ll[len(got_ll):].any()
        assert got_d == [int(x) for x in d[:len(got_d)]] and not d[len(got_d):].any()
