#!/usr/bin/env python3
"""bench.py — Mreads/s through coordinate sort + mark duplicates (+ optical metrics) + BQSR gather + finalize + apply.

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run, one rank per GPU.  A "step" is one pass of the whole hot path over the rank's resident shard of
synthetic 150 bp paired-end reads (inputs are in HBM before the timed region starts; staging over PCIe is reported
separately and is never `value`).  Rank 0 prints ONE JSON line.

Multi-GPU (weak scaling): every rank holds its own shard (`--reads` per GPU, contig-partition style, no data-path
collective); the only exchange is one RCCL all-reduce of the BQSR count tables + duplicate metrics (the sfm
"sum the per-split tables" step, cmd/sfm.go:769-805 / filters/print-bqsr.go:310-329), after which every rank finalizes
and applies locally.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per read, SURVEY.md §8(d) / BASELINE.md §3
BYTES_PER_READ = {"adapt": 185, "sort": 152, "markdup": 138, "bqsr_gather": 424, "bqsr_apply": 385}
BYTES_FULL_PATH = 1284
HBM_PEAK_GBS = 8000.0
MAX_CYCLE = 500


def kernel_stage(name: str) -> str:
    if name.startswith("adapt"):
        return "adapt"
    if name.startswith("md_"):
        return "markdup"
    if name.startswith("mx_"):
        return "metrics"
    if name.startswith("bqsr_apply"):
        return "bqsr_apply"
    if name.startswith("bqsr_"):
        return "bqsr_gather"
    return "sort"  # radix_*, scan_*, tie_*, iota, material_*, seg_*, large_*, add_own


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ELP_BENCH_READS", 50_000_000)), help="reads per GPU (approximate: pairs = reads/2)")
    ap.add_argument("--genome", default="c3", help="synthetic genome preset (tools/synth): c3 = hg38/12, 24 contigs")
    ap.add_argument("--cpu-reads", type=int, default=2_000_000, help="sample size for the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from elprep_amd.engine import BqsrTables, Engine
    from tools import synth

    cfg = synth.config(args.genome)
    hdr = cfg.header()
    pairs_per_rank = args.reads // 2
    p_lo, p_hi = rank * pairs_per_rank, (rank + 1) * pairs_per_rank

    # ---- generate the shard and stage it (untimed; PCIe-inclusive staging rate reported separately)
    eng = Engine(hdr, local_rank if world > 1 else 0)
    t0 = time.time()
    chunk = 1_000_000
    n_total = 0
    stage_s = 0.0
    qual_bytes = 0
    # the generator is deterministic per pair index, so chunks are produced by a small thread pool (ctypes releases the
    # GIL) and staged strictly in order
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(12, (os.cpu_count() or 2) // max(world, 1)))
    ranges = [(lo, min(lo + chunk, p_hi)) for lo in range(p_lo, p_hi, chunk)]
    with ThreadPoolExecutor(workers) as pool:
        q = deque()
        it = iter(ranges)
        for _ in range(workers + 1):
            r = next(it, None)
            if r is not None:
                q.append(pool.submit(synth.generate, cfg, r[0], r[1]))
        while q:
            b = q.popleft().result()
            r = next(it, None)
            if r is not None:
                q.append(pool.submit(synth.generate, cfg, r[0], r[1]))
            ts = time.time()
            eng.stage(b)
            stage_s += time.time() - ts
            n_total += b.n
            qual_bytes += int(b.qual_off[-1])
            del b
    gen_s = time.time() - t0 - stage_s
    for r in range(hdr.n_ref):
        eng.set_reference(r, synth.reference(cfg, r))
        eng.set_known_sites(r, flatten_sites(synth.known_sites_raw(cfg, r)))
    eng.sync()

    eng.snapshot()  # FLAG and QUAL are the only columns the path mutates; every step starts from the same staged input

    def restore():
        eng.rollback()

    def step():
        eng.sort_coordinate(fetch=False)
        eng.mark_duplicates(True, fetch=False)
        ctr = eng.dup_metrics(100)
        qt, ct, xt = eng.recalibrate(MAX_CYCLE)
        if world > 1:  # single all-reduce of tables + metrics over RCCL/xGMI
            flat = torch.from_numpy(np.concatenate([qt.ravel(), ct.ravel(), xt.ravel(), ctr.ravel()])).to(dev)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat = flat.cpu().numpy()
            a, b_, c_ = qt.size, ct.size, xt.size
            qt, ct, xt = flat[:a].reshape(qt.shape), flat[a:a + b_].reshape(ct.shape), flat[a + b_:a + b_ + c_].reshape(xt.shape)
        tb = BqsrTables(qt, ct, xt, MAX_CYCLE).finalize()
        lut, present = tb.build_lut(0)
        eng.apply_bqsr(lut, present, MAX_CYCLE, fetch=False)
        eng.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        restore()
        step()
    eng.profile_enable(True)
    eng.profile_reset()
    restore_s = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr = time.perf_counter()
        restore()
        eng.sync()
        restore_s += time.perf_counter() - tr
        step()
    barrier()
    elapsed = time.perf_counter() - t0 - restore_s  # state restore (two D2D copies) is bookkeeping, not part of the path
    prof = eng.profile()
    eng.profile_enable(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nt = torch.tensor([n_total], dtype=torch.int64, device=dev)
        dist.all_reduce(nt, op=dist.ReduceOp.SUM)
        n_global = int(nt.item())
    else:
        n_global = n_total

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_global / (elapsed / args.steps) / 1e6
        # per-stage kernel time (HIP events on the ctx stream) -> dominant kernel roofline
        stage_ms = {}
        for name, (cnt, ms) in prof.items():
            st = kernel_stage(name)
            stage_ms[st] = stage_ms.get(st, 0.0) + ms / args.steps
        kern_ms = {name: ms / max(cnt, 1) for name, (cnt, ms) in prof.items()}
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
        dom_stage = kernel_stage(dom)
        dom_launch_ms = kern_ms[dom]
        dom_bytes = BYTES_PER_READ.get(dom_stage, 0) * n_total
        launches_per_step = prof[dom][0] / args.steps
        # achieved = algorithmic bytes of the stage the dominant kernel belongs to, per launch of that kernel
        achieved = (dom_bytes / max(launches_per_step, 1)) / (dom_launch_ms * 1e-3) / 1e9 if dom_launch_ms > 0 else 0.0
        kernel_total_ms = sum(stage_ms.values())
        out = {
            "metric": "Mreads/s through sort+markdup+BQSR, 150bp PE",
            "value": round(value, 3),
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32/int64 (integer path; float64 finalize on host)",
            "data": "synthetic",
            "config": {"workload": f"C3-style: {n_total} reads/GPU 150bp PE, genome {args.genome} (24 contigs hg38/12), sort+markdup+optical metrics+BQSR gather+finalize+apply",
                       "reads_per_gpu": n_total, "max_cycle": MAX_CYCLE, "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "kernel": dom, "stage": dom_stage, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "path_frac": round((BYTES_FULL_PATH * n_total / (kernel_total_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 5) if kernel_total_ms else None},
            "stage_ms_per_step": {k: round(v, 3) for k, v in sorted(stage_ms.items())},
            "kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]},
            "staging": {"gen_s": round(gen_s, 2), "h2d_stage_s": round(stage_s, 2)},
        }
        if not args.no_cpu_baseline and args.cpu_reads > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, hdr, args.cpu_reads)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def flatten_sites(raw: np.ndarray) -> np.ndarray:
    """intervals.ParallelSortByStart + ParallelFlatten on the host (intervals/intervals.go:77-132) — vectorised numpy."""
    if raw.shape[0] == 0:
        return raw.astype(np.int32)
    o = np.argsort(raw[:, 0], kind="stable")
    s, e = raw[o, 0].astype(np.int64), raw[o, 1].astype(np.int64)
    run_end = np.maximum.accumulate(e)
    new = np.ones(s.size, dtype=bool)
    new[1:] = s[1:] > run_end[:-1]
    starts = s[new]
    grp = np.cumsum(new) - 1
    ends = np.zeros(starts.size, dtype=np.int64)
    np.maximum.at(ends, grp, e)
    return np.stack([starts, ends], axis=1).astype(np.int32)


def cpu_baseline(cfg, hdr, n_reads):
    """The CPU oracle (plain-C restatement of the reference's algorithms, single thread) timed on a bounded sample of the
    same workload on this box's host cores.  kind = "port": it is NOT the elPrep binary (no Go toolchain)."""
    import oracle as orc
    from tools import synth
    b = synth.generate(cfg, 0, n_reads // 2)
    refs = [synth.reference(cfg, r) for r in range(hdr.n_ref)]
    sites = [flatten_sites(synth.known_sites_raw(cfg, r)) for r in range(hdr.n_ref)]
    t0 = time.perf_counter()
    perm = orc.sort_coordinate(b)
    flags, ctr, _ = orc.dup_metrics(b, hdr, perm, 100)
    qt, ct, xt = orc.bqsr_gather(b, hdr, orc.BqsrRef(refs, sites), flags, MAX_CYCLE)
    fin = orc.BqsrFinal(qt, ct, xt, MAX_CYCLE)
    fin.apply(b, hdr, 0)
    dt = time.perf_counter() - t0
    return {"value": round(b.n / dt / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": f"{b.n} reads of the same synthetic workload, full path (sort+markdup+optical metrics+BQSR gather+finalize+apply), "
                      f"single-threaded C restatement of the reference algorithms (oracle/), {dt:.1f} s; host has {os.cpu_count()} cores"}


if __name__ == "__main__":
    main()
