#!/usr/bin/env python3
"""Inputs for a run of the REAL elPrep on the synthetic reads of the parity tests (tools/ref/make_fixtures.sh): in.sam, ref.fasta,
sites.bed and case.json (how to regenerate the same reads).  Everything is a function of (genome preset, seed index, pairs) through
tools/synth, so tests/test_oracle_golden.py rebuilds the identical batch and compares the oracle with what elPrep wrote.
usage: write_inputs.py <outdir> [pairs] [seed_index]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

NIB = "=ACMGRSVTWYHKDBN"
OPS = "MIDNSHP=X"


def sam_line(b, i, h, names):
    flag = int(b.flag[i])
    rname = names[b.refid[i]] if b.refid[i] >= 0 else "*"
    if b.next_refid[i] < 0:
        rnext = "*"
    elif b.next_refid[i] == b.refid[i]:
        rnext = "="
    else:
        rnext = names[b.next_refid[i]]
    cig = b.cigar[int(b.cigar_off[i]):int(b.cigar_off[i + 1])]
    cigar = "".join(f"{int(c) >> 4}{OPS[int(c) & 15]}" for c in cig) or "*"
    s4 = b.seq4[int(b.seq_off[i]):int(b.seq_off[i + 1])]
    n = int(b.l_seq[i])
    seq = "".join(NIB[(int(s4[k >> 1]) >> (0 if k & 1 else 4)) & 15] for k in range(n)) or "*"
    q = b.qual[int(b.qual_off[i]):int(b.qual_off[i + 1])]
    qual = "".join(chr(int(x) + 33) for x in q) or "*"
    fields = [b.qname_of(i).decode(), str(flag), rname, str(int(b.pos[i])), str(int(b.mapq[i])), cigar, rnext, str(int(b.pnext[i])), str(int(b.tlen[i])),
              seq, qual]
    if b.rgid[i] != 0xFFFF:
        fields.append("RG:Z:" + h.rg_ids[int(b.rgid[i])])
    return "\t".join(fields)


def main():
    out = sys.argv[1]
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seed_index = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    os.makedirs(out, exist_ok=True)
    cfg = synth.config("tiny", seed_index)
    cfg.p_frag = 0.02
    b = synth.generate(cfg, 0, pairs)
    h = cfg.header()
    names = cfg.ref_names
    with open(os.path.join(out, "in.sam"), "w") as f:
        f.write("@HD\tVN:1.6\tSO:unknown\n")
        for nm, ln in zip(names, cfg.ref_len):
            f.write(f"@SQ\tSN:{nm}\tLN:{ln}\n")
        for lane in range(1, cfg.n_lanes + 1):
            half = (cfg.n_lanes + 1) // 2
            f.write(f"@RG\tID:rg{lane}\tLB:{'lib1' if (lane - 1) < half else 'lib2'}\tPU:FC1.{lane}\tSM:s1\tPL:illumina\n")
        for i in range(b.n):
            f.write(sam_line(b, i, h, names) + "\n")
    with open(os.path.join(out, "ref.fasta"), "w") as f:
        for r, nm in enumerate(names):
            seq = synth.reference(cfg, r).tobytes().decode()
            f.write(f">{nm}\n")
            for k in range(0, len(seq), 60):
                f.write(seq[k:k + 60] + "\n")
    with open(os.path.join(out, "sites.bed"), "w") as f:  # BED: 0-based start, end exclusive <- 1-based inclusive intervals
        for r, nm in enumerate(names):
            for s, e in synth.known_sites_raw(cfg, r):
                f.write(f"{nm}\t{int(s) - 1}\t{int(e)}\n")
    json.dump({"genome": "tiny", "seed_index": seed_index, "pairs": pairs, "p_frag": 0.02, "records": int(b.n)}, open(os.path.join(out, "case.json"), "w"))
    print(f"wrote {b.n} records to {out}")


if __name__ == "__main__":
    main()
