"""Shared fixtures for the parity tests: deterministic synthetic batches + oracle-side BQSR inputs."""
import functools

import numpy as np

import oracle as orc
from tools import synth


@functools.lru_cache(maxsize=8)
def dataset(name: str, n_pairs: int, seed_index: int = 0, p_frag: float = 0.0):
    cfg = synth.config(name, seed_index)
    cfg.p_frag = p_frag
    b = synth.generate(cfg, 0, n_pairs)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    return cfg, b, h, refs, sites
