#!/usr/bin/env python3
"""bench.py — Mreads/s through mark duplicates + coordinate sort (+ optical metrics) + BQSR gather + finalize + apply.

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run, one rank per GPU.  A "step" is one pass of the whole hot path over the rank's resident shard of
synthetic 150 bp paired-end reads (inputs are in HBM before the timed region starts; staging over PCIe is reported
separately and is never `value`).  Rank 0 prints ONE JSON line.

N = 1 is `elprep filter` (one context): BASELINE.json's config C3 (50 M reads, sort + markdup + BQSR), the largest single-GPU
configuration.  The same line carries, as `extra`, the other single-GPU facts asked for: config C2 (mark duplicates + sort only) on
the same staged reads, the path on a read set with ~40 distinct quality values (the 7-value binned qualities of the main workload
are the easy case for the BQSR kernels), and the PCIe-inclusive staging rate through elp_stage_bam.

N > 1 is `elprep sfm`: contig groups -> ranks; the only exchange per step is ONE all-reduce (RCCL through the C ABI's device
group) of the BQSR count tables + duplication counters.  `--scaling weak` (default; per-GPU reads fixed) or `--scaling strong
--total-reads R` (a fixed read set partitioned by the real computeContigGroups of the genome: per-rank counts and the imbalance
are in the line).
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


def effective_cores() -> int:
    """host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants a container 16 CPUs' worth of time: cpu.max = 1600000 100000)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# The synthetic generator and the CPU baseline are OpenMP code: without these two settings every parallel region starts one thread
# per VISIBLE CPU (256 on the GPU box, of which the container may use 16) and the idle ones spin - which starves the host thread
# that finalises the BQSR tables inside the timed steps (measured: 15 ms of host time per step instead of 2).
os.environ.setdefault("OMP_NUM_THREADS", str(effective_cores()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

# algorithmic HBM bytes per read, SURVEY.md §8(d) / BASELINE.md §3
BYTES_PER_READ = {"adapt": 185, "sort": 152, "markdup": 138, "bqsr_gather": 424, "bqsr_apply": 385}
BYTES_FULL_PATH = 1284
BYTES_C2 = 475
HBM_PEAK_GBS = 8000.0
MAX_CYCLE = 500


def kernel_stage(name: str) -> str:
    if name.startswith("adapt") or name.startswith("flat_index") or name.startswith("qual_present"):
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


def timed(step, restore, steps, warmup, prof_eng, barrier):
    """W untimed + K timed steps, bracketed by barrier + device sync; restore (two D2D copies) is bookkeeping, not the path"""
    for _ in range(warmup):
        restore()
        step()
    prof_eng.profile_enable(True)
    prof_eng.profile_reset()
    restore_s = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr = time.perf_counter()
        restore()
        restore_s += time.perf_counter() - tr
        step()
    barrier()
    elapsed = time.perf_counter() - t0 - restore_s
    prof = prof_eng.profile()
    prof_eng.profile_enable(False)
    return elapsed, prof


def summarize(prof, steps, n_reads, full_bytes, wall_ms=None, concurrent=False, dom=None):
    """per-stage kernel ms per step, the dominant kernel and its roofline figures.  `achieved` = the algorithmic bytes of the stage the
    dominant kernel belongs to (SURVEY.md 8d x the reads of the step) over the time ALL launches of that kernel take in one step (a
    kernel launched several times per step - the radix scatter - is not priced by the average of its unequal launches); `stage_frac` =
    the same per stage, over the stage's whole kernel time."""
    stage_ms = {}
    for name, (cnt, ms) in prof.items():
        st = kernel_stage(name)
        stage_ms[st] = stage_ms.get(st, 0.0) + ms / steps
    # round 6: mark duplicates' front pass (md_front) also does the adapt stage's fixed-field part (unclipped positions, sort keys: the
    # 35 bytes per read SURVEY 8(d) budgets under "adapt" next to the 150 QUAL bytes of the score): its time is booked under markdup, so
    # the bytes go there too - the stages' sum, and the path's, do not change
    bpr = dict(BYTES_PER_READ)
    if any(k.endswith("md_front") for k in prof):
        bpr["adapt"], bpr["markdup"] = BYTES_PER_READ["adapt"] - 35, BYTES_PER_READ["markdup"] + 35
    # the dominant kernel: the one with the most time per step - taken from the serial-order steps (`dom`) when this profile's stages ran
    # at once: a kernel that WAITS for another stream's kernel (the sort's passes under the count kernel) is not the one that dominates
    if dom is None or dom not in prof:
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
    dom_stage = kernel_stage(dom)
    launches_per_step = max(prof[dom][0] / steps, 1)
    dom_ms_per_step = prof[dom][1] / steps
    achieved = (bpr.get(dom_stage, 0) * n_reads) / (dom_ms_per_step * 1e-3) / 1e9 if dom_ms_per_step > 0 else 0.0
    kernel_total_ms = sum(stage_ms.values())
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if dom in tj["bytes_per_read"]:
            traffic = round(tj["bytes_per_read"][dom] * n_reads)
    except Exception:
        traffic = None
    stage_frac = {st: round(bpr[st] * n_reads / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for st, ms in sorted(stage_ms.items())
                  if st in bpr and ms > 0}
    if "adapt" in stage_ms and "markdup" in stage_ms:  # the two stages the front pass joins, as one figure whatever the booking
        both = stage_ms["adapt"] + stage_ms["markdup"]
        stage_frac["adapt+markdup"] = round((BYTES_PER_READ["adapt"] + BYTES_PER_READ["markdup"]) * n_reads / (both * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roof = {"bound": "hbm", "kernel": dom, "stage": dom_stage, "launches_per_step": launches_per_step, "kernel_ms_per_step": round(dom_ms_per_step, 4),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            # the same stage's algorithmic bytes over ALL its kernels (the prologues that feed the dominant kernel included): the figure `frac`
            # flatters, next to it on purpose
            "frac_of_whole_stage": stage_frac.get(dom_stage), "traffic": traffic,
            "traffic_source": "profiles/traffic.json: rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE) of an 8 M-read run of this path, per read, scaled to this run's reads - not measured in this run",
            "stage_frac": stage_frac, "stage_bytes_per_read": bpr,
            "path_frac": round((full_bytes * n_reads / (kernel_total_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 5) if kernel_total_ms else None}
    if wall_ms:
        roof["path_frac_wall"] = round((full_bytes * n_reads / (wall_ms * 1e-3) / 1e9) / HBM_PEAK_GBS, 5)
    if concurrent:
        # the stages run at once on three streams (sort, metrics, BQSR chain): every kernel's event time includes what it waited for a
        # CU, LDS or bandwidth another stream's kernel held - the sums exceed the step and the per-stage fractions are those of CONTENDED
        # kernels; the path's figure is the wall one, the uncontended stage table is under `serial_order`
        roof["path_frac"] = roof.get("path_frac_wall")
        roof["concurrent_streams"] = ("sort, duplication metrics and the BQSR chain run at once behind mark duplicates: kernel times are contended, "
                                      "their sum exceeds the step; `path_frac` is over the step's wall time; uncontended kernels: `serial_order`")
    kern = {k: round(v[1] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:28]}
    return {k: round(v, 3) for k, v in sorted(stage_ms.items())}, kern, roof


def main():
    # the ONE line on stdout is the result: whatever libraries print there (RCCL greets with five lines of versions when its first
    # communicator comes up) goes to stderr - file descriptor 1 points at stderr until the line is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ELP_BENCH_READS", 50_000_000)), help="reads per GPU (approximate: pairs = reads/2)")
    ap.add_argument("--genome", default="c3", help="synthetic genome preset (tools/synth): c3 = hg38/12, 24 contigs")
    ap.add_argument("--quals", choices=["binned", "full"], default="binned", help="quality alphabet of the main workload: 7 binned values or ~40 values")
    ap.add_argument("--stages", choices=["full", "c2"], default="full", help="full = BASELINE config C3 (sort+markdup+BQSR); c2 = mark duplicates + sort only")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N > 1: per-GPU reads fixed, or --total-reads partitioned by contig groups")
    ap.add_argument("--total-reads", type=int, default=0, help="strong scaling: reads of the whole job (default: --reads)")
    ap.add_argument("--cpu-reads", type=int, default=8_000_000, help="sample size for the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the device path with the CPU leg's outputs (it needs the CPU leg)")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the C2 / full-quality / PCIe-inclusive side measurements")
    ap.add_argument("--extra-reads", type=int, default=16_000_000, help="reads of the full-quality side measurement")
    ap.add_argument("--c4-reads", type=int, default=int(os.environ.get("ELP_BENCH_C4_READS", 75_000_000)),
                    help="N = 1 side measurement `c4_share`: reads on the hg38-sized genome (0 = skip)")
    ap.add_argument("--mode", choices=["auto", "filter", "sfm"], default="auto",
                    help="auto: `elprep filter` (one context) when started as a plain process, the `elprep sfm` step (contig-group splits + spread split, "
                         "all-reduce) whenever started under torch.distributed.run - also with ONE rank, so that a scaling curve's N = 1 point runs the "
                         "same code as its N > 1 points")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ELP_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a box with fewer GPUs than ranks (ranks then share devices and
    # the collectives run over gloo on the host); the driver's runs use the default: nccl = RCCL over xGMI, one GPU per rank
    backend = os.environ.get("ELP_BENCH_BACKEND", "nccl")
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    sfm_mode = world > 1 or args.mode == "sfm" or (args.mode == "auto" and under_launcher)
    if sfm_mode and not under_launcher:
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        # one process per GPU on one node: the host's cores are shared.  The table path's worker pool takes its share (all ranks finalise
        # at the same moment, right behind the all-reduce), and a rank only polls its stream while it waits (ELP_SYNC_SPIN) if every
        # rank's two waiting threads can keep a core of their own next to that
        share = max(1, effective_cores() // world)
        os.environ.setdefault("ELP_HOST_THREADS", str(share))
        if share < 4:
            os.environ.setdefault("ELP_SYNC_SPIN", "0")
    if sfm_mode:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if sfm_mode else 0)
    cdev = dev if (not sfm_mode or backend == "nccl") else torch.device("cpu")  # where collective buffers live

    from elprep_amd.engine import BqsrTables, Engine
    from tools import synth

    cfg = synth.config(args.genome)
    cfg.qual_mode = 1 if args.quals == "full" else 0
    hdr = cfg.header()
    pairs_per_rank = args.reads // 2
    refs_sites = [(r, synth.reference(cfg, r), flatten_sites(synth.known_sites_raw(cfg, r))) for r in range(hdr.n_ref)]

    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(12, effective_cores() // max(world, 1)))
    chunk = 1_000_000
    host_pool = ThreadPoolExecutor(1)
    metrics_pool = ThreadPoolExecutor(1)
    sort_pool = ThreadPoolExecutor(1)
    # how a step's stages are driven (the library runs the sort and the metrics pass on side lanes of the context; the host decides what
    # it calls at once): "three" (default) = sort, metrics and the BQSR chain all at once behind mark duplicates - the fastest step; every
    # kernel's time then includes what it waited for another stream's kernel, so the line also carries the same step's kernels measured
    # one after the other (`serial_order`, outside the timed region); "metrics" = the metrics pass under the sort, both behind the gather
    # (the BQSR kernels have the GPU to themselves); "serial" = one after the other, as until round 5
    # "auto" (default): three steps of "three" and of "metrics" each in front of the warm-up, the faster order is the one that is timed - which
    # of the two wins is a property of the box's HOST (how well three threads' launches interleave), not of the kernels: one build measured
    # 20.57 / 20.88 ms on one box of the pool and 21.98 / 21.55 on another (profiles/round6_*, round6d_*)
    order = os.environ.get("ELP_BENCH_ORDER", "auto")
    order_probe = None
    # elp_sort_ahead (the sort's key passes queued from inside mark duplicates): measured on the bench's step, it does not pay - the passes'
    # 1024-thread workgroups find no CU while the pair phase's and the prologues' small workgroups keep every CU partly occupied, and with
    # smaller tiles they slow the pair phase by what they gain (profiles/round6_sort_ahead_ab.txt); off unless asked for
    sort_ahead = os.environ.get("ELP_BENCH_SORT_AHEAD", "0") != "0"

    def generated(jobs):
        """yield the batches of `jobs` = [(config, pair_lo, pair_hi), ...] in order; the generator is deterministic per pair index,
        so chunks are produced by a small thread pool (ctypes releases the GIL)"""
        with ThreadPoolExecutor(workers) as pool:
            q = deque()
            it = iter(jobs)
            for _ in range(workers + 1):
                j = next(it, None)
                if j is not None:
                    q.append(pool.submit(synth.generate, *j))
            while q:
                b = q.popleft().result()
                j = next(it, None)
                if j is not None:
                    q.append(pool.submit(synth.generate, *j))
                yield b

    def barrier():
        if sfm_mode:
            dist.barrier()
        torch.cuda.synchronize()

    host_parts = []  # per step: [tables to the host, FinalizeBQSRTables, LUT rows, LUT upload] in ms (ELP_BENCH_HOST_PARTS=1 prints them per side run)
    host_ms, wait_ms = [], []  # per step: the host's FinalizeBQSRTables + LUT (ms), and the part of it the device's sort + metrics did not hide

    def make_filter_steps(eng, lut_buf):
        """the step functions of one `elprep filter` context.  Order of events as in the reference (cmd/filter.go:142-211): MarkDuplicates
        is a filter of the phase-1 pipeline, the sort is that pipeline's Finalize (sam/filter-pipeline.go:116), then the metrics pass,
        Recalibrate, finalize, ApplyBQSR."""
        def finalize_lut():
            """the tables' way to the host, FinalizeBQSRTables, the LUT and its way back - in the ROWS form (round 5): only the rows of the
            qualities the gather gave table slots cross PCIe and are scanned / filled by the host (with 16 read groups 1.8 MB of tables
            down and 1.9 MB of LUT up instead of 24 + 26 MB); the dense forms if the tables hold other rows (summed tables)"""
            t0 = time.perf_counter()
            quals = eng.quals_counted()
            got = eng.tables_fetch_rows(quals, reuse=True)
            t1 = time.perf_counter()
            if got is not None:
                tb = BqsrTables.from_rows(eng.header.n_cov, quals, *got, MAX_CYCLE).finalize()
                t2 = time.perf_counter()
                if lut_buf[0] is None or lut_buf[0][0] != tuple(quals):  # built in page-locked memory (its upload runs at the PCIe rate), once per context
                    lut_buf[0] = (tuple(quals), (eng.pinned_zeros((eng.header.n_cov, len(quals), 2 * MAX_CYCLE + 1, 17), np.uint8),
                                                 np.zeros((eng.header.n_cov, 94), np.uint8), np.zeros(eng.header.n_cov, np.uint8)))
                rows, defaults, present = tb.build_lut_rows(quals, 0, out=lut_buf[0][1])
                t3 = time.perf_counter()
                eng.lut_upload_rows(quals, rows, defaults, present, MAX_CYCLE)  # on the context's copy stream, from this (host) thread, while the GPU sorts
                host_parts.append([round((b - a) * 1e3, 3) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, time.perf_counter()))])
            else:
                tb = BqsrTables(*eng.tables_fetch(reuse=True), MAX_CYCLE).finalize()
                lut, present = tb.build_lut(0)
                eng.lut_upload(lut, present, MAX_CYCLE)
            host_ms.append((time.perf_counter() - t0) * 1e3)

        def step_full(order=order):
            eng.sort_ahead(order == "three" and sort_ahead)  # (the sort's key passes queued from inside mark duplicates, elp_sort_ahead)
            eng.mark_duplicates(True, fetch=False)
            if order == "three":
                # round 6: behind mark duplicates three chains need nothing of each other - the coordinate sort (the pipeline's Finalize,
                # sam/filter-pipeline.go:116), the duplication-metrics pass, and Recalibrate -> FinalizeBQSRTables -> ApplyBQSR - and the
                # library runs the first two on side lanes of the context (streams, scratch and error words of their own:
                # elp_sort_coordinate, elp_dup_metrics), so a host thread each drives them while this one drives the BQSR chain.  The
                # reference runs them one after the other (cmd/filter.go:162-196); what the step does is the same.
                st = sort_pool.submit(eng.sort_coordinate, False)
                mx = metrics_pool.submit(eng.dup_metrics, 100)
                eng.recalibrate_device(MAX_CYCLE)
                finalize_lut()
                wait_ms.append(0.0)
                eng.apply_bqsr(None, None, MAX_CYCLE, fetch=False)
                st.result()
                mx.result()
                eng.sync()
                return
            eng.recalibrate_device(MAX_CYCLE)
            # the tables' way to the host, the float64 finalisation + LUT (host thread; the copy runs on the context's copy stream) on one
            # side, the coordinate sort and the duplication-metrics pass (device, this thread) on the other do not depend on each other:
            # the host finalises while the GPU sorts and counts (the reference runs them one after the other, cmd/filter.go:162-196;
            # its sort is the pipeline's Finalize and needs nothing of BQSR either)
            fin = host_pool.submit(finalize_lut)
            if order == "serial":
                eng.sort_coordinate(fetch=False)
                eng.dup_metrics(100)
            else:
                # the duplication-metrics pass on its side lane, driven by a second host thread while this one drives the sort
                mx = metrics_pool.submit(eng.dup_metrics, 100)
                eng.sort_coordinate(fetch=False)
                mx.result()
            t_dev = time.perf_counter()
            fin.result()
            wait_ms.append((time.perf_counter() - t_dev) * 1e3)  # what the device's sort + metrics did not hide of the host's finalisation
            eng.apply_bqsr(None, None, MAX_CYCLE, fetch=False)
            eng.sync()

        def step_c2():
            eng.mark_duplicates(True, fetch=False)
            eng.sort_coordinate(fetch=False)
            eng.sync()

        def restore():
            eng.rollback()
            eng.sync()
        return step_full, step_c2, restore

    t0 = time.time()
    stage_s = 0.0
    route_s = 0.0  # sfm: classifying the generated records by split and exchanging the few that belong to another rank
    dev_id = local_rank if sfm_mode else 0
    per_rank_reads = None
    if not sfm_mode:
        # ---- `elprep filter`: one context holds everything (untimed staging; PCIe-inclusive rate reported separately)
        eng = Engine(hdr, dev_id)
        n_total = 0
        for b in generated([(cfg, lo, min(lo + chunk, pairs_per_rank)) for lo in range(0, pairs_per_rank, chunk)]):
            ts = time.time()
            eng.stage(b)
            stage_s += time.time() - ts
            n_total += b.n
            del b
        for r, ref, sites in refs_sites:
            eng.set_reference(r, ref)
            eng.set_known_sites(r, sites)
        eng.sync()
        eng.snapshot()  # FLAG and QUAL are the only columns the path mutates; every step starts from the same staged input
        prof_eng = eng
        step_full, step_c2, restore = make_filter_steps(eng, [None])
        if order == "auto":
            order = "three"
            if args.stages == "full":
                order_probe = {}
                for o in ("three", "metrics"):
                    elo, _ = timed(lambda: step_full(o), restore, 3, 1, eng, barrier)
                    order_probe[o] = round(elo / 3 * 1e3, 3)
                order = min(order_probe, key=order_probe.get)
        step = (lambda: step_full(order)) if args.stages == "full" else step_c2
        mode = "filter"
    else:
        # ---- `elprep sfm`: contig groups -> ranks; every rank produces the reads of the groups it owns, the few records that
        # belong to another rank's split (spread mates, supplementary alignments, unmapped pairs) are routed point to point;
        # per step ONE all-reduce (RCCL over xGMI, through the C ABI's device group) of the BQSR count tables + duplication counters
        from elprep_amd import sfm
        comm = sfm.Comm(cdev)
        gof, G = sfm.contig_groups(cfg.ref_len)  # computeContigGroups on the genome's @SQ lengths (hg38 proportions)
        ranges = sfm.group_ranges(gof, G)
        glen = [float(sum(cfg.ref_len[lo:hi])) for lo, hi in ranges]
        weights = [0.012 * sum(glen)] + glen + [0.03 * sum(glen)]
        owner = sfm.assign_splits(weights, world)
        mine = [g for g in range(1, G + 1) if owner[g] == rank]
        mylen = sum(glen[g - 1] for g in mine)
        jobs = []
        if args.scaling == "weak":
            # per-GPU work is fixed: every rank ends up with about `--reads` records.  The owners of the spread and of the unmapped
            # split receive records from everybody (fractions measured on a small sample), so they generate fewer of their own.
            sample = synth.generate(cfg, 0, 20000)
            sg, ssp = sfm.split_records(sample, gof)
            f_spread, f_unmapped = float(ssp.mean()), float((sg == 0).mean())
            extra = (f_spread * world if owner[G + 1] == rank else 0.0) + (f_unmapped * world if owner[0] == rank else 0.0) - f_unmapped
            my_pairs = max(int(pairs_per_rank * (1.0 - extra)), pairs_per_rank // 4)
            pairs_of_group = {g: int(my_pairs * glen[g - 1] / max(mylen, 1.0)) for g in mine}
        else:
            # total work is fixed: the read set of the whole job is spread over the genome, a group gets its share by length, a rank
            # the groups computeContigGroups + the balanced assignment give it - the imbalance of the real `sfm` split shows
            total_pairs = (args.total_reads or args.reads) // 2
            pairs_of_group = {g: int(total_pairs * glen[g - 1] / sum(glen)) for g in mine}
        for g in mine:
            c = synth.config(args.genome)
            c.qual_mode = cfg.qual_mode
            # a seed of its own per group (distinct read names and layouts), ONE genome: the reference sequence and the known sites keep the
            # main configuration's seed (rounds 2 and 3 let them follow the group's seed: the reads were reads of another genome, three
            # bases in four a mismatch, and the sfm path was timed on that)
            c.seed = cfg.seed + 7919 * g
            c.ref_seed = cfg.seed
            c.home_lo, c.home_hi = ranges[g - 1]
            jobs += [(c, lo, min(lo + chunk, pairs_of_group[g])) for lo in range(0, pairs_of_group[g], chunk)]
        rounds = torch.tensor([len(jobs)], dtype=torch.int64, device=cdev)
        dist.all_reduce(rounds, op=dist.ReduceOp.MAX)  # every rank takes part in every routing round
        rk = sfm.SfmRank(hdr, dev_id, comm)
        n_total = 0
        gen = generated(jobs)
        for _ in range(int(rounds.item())):
            b = next(gen, None)
            tr = time.time()
            # the split phase through the C ABI: staged into the rank's reader context, classified on the device (elp_split_classify), every
            # record delivered device to device to the context of its split's owner (elp_copy_records / elp_exchange_records: RCCL send /
            # receive over xGMI when every rank has its own GPU)
            if os.environ.get("ELP_SFM_ROUTE") == "host":  # escape hatch: round 4's routing through host numpy + torch.distributed
                got = sfm.route(b if b is not None else sfm.empty_batch(), gof, G, owner, comm)
                rk.stage(0, got.local)
                rk.stage(1, got.spread)
                del got
            else:
                rk.route(b if b is not None else sfm.empty_batch(), gof, G, owner)
            route_s += time.time() - tr
            del b
        n_total = rk.n_reads  # the tagged copies are not reads of their own
        for r, ref, sites in refs_sites:
            rk.set_reference(r, ref)
            rk.set_known_sites(r, sites)
        rk.sync()
        rk.snapshot()
        prof_eng = rk.engines[0]
        lut_buf = [None]

        def restore():
            rk.rollback()
            rk.sync()

        def finalize_lut(qt, ct, xt):
            tb = BqsrTables(qt, ct, xt, MAX_CYCLE).finalize()
            lut, present = tb.build_lut(0, out=lut_buf[0])
            lut_buf[0] = (lut, present)
            return lut, present

        rows_buf = [None]

        def finalize_lut_rows(quals, q_rows, c_rows, x_rows):
            # the rows form of the filter step's table path (only the rows of the counted qualities cross PCIe; the LUT rows are built in
            # page-locked memory once per set of qualities)
            tb = BqsrTables.from_rows(hdr.n_cov, quals, q_rows, c_rows, x_rows, MAX_CYCLE).finalize()
            if rows_buf[0] is None or rows_buf[0][0] != tuple(quals):
                e0 = rk.engines[0]
                rows_buf[0] = (tuple(quals), (e0.pinned_zeros((hdr.n_cov, len(quals), 2 * MAX_CYCLE + 1, 17), np.uint8),
                                              np.zeros((hdr.n_cov, 94), np.uint8), np.zeros(hdr.n_cov, np.uint8)))
            return tb.build_lut_rows(quals, 0, out=rows_buf[0][1])

        def step():
            # same shape as the filter step: the host finalises the all-reduced tables (and uploads the LUT to both contexts) while the
            # GPU sorts the rank's splits
            rk.step(MAX_CYCLE, 100, host_pool, finalize_lut, None if os.environ.get("ELP_SFM_DENSE_TABLES") else finalize_lut_rows)
            rk.sync()
        mode = "sfm"
        counts = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_total], dtype=torch.int64, device=cdev))
        per_rank_reads = [int(c.item()) for c in counts]
    gen_s = time.time() - t0 - stage_s - route_s

    elapsed, prof = timed(step, restore, args.steps, args.warmup, prof_eng, barrier)

    if sfm_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n_global = sum(per_rank_reads)
    else:
        n_global = n_total

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_global / (elapsed / args.steps) / 1e6
        full_bytes = BYTES_FULL_PATH if args.stages == "full" else BYTES_C2
        conc = (not sfm_mode and args.stages == "full" and order == "three") or sfm_mode
        serial_order = None
        dom_serial = None
        if not sfm_mode and args.stages == "full" and order != "serial":
            # the same step with its stages one after the other (as until round 5), OUTSIDE the timed region: what every kernel takes when it
            # has the GPU to itself - the stage table and the dominant kernel's roofline without the other streams' kernels in its way
            el_s, prof_s = timed(lambda: step_full("serial"), restore, 3, 1, eng, barrier)
            st_s, km_s, rf_s = summarize(prof_s, 3, n_total, full_bytes, el_s / 3 * 1e3, False)
            dom_serial = rf_s["kernel"]
            serial_order = {"ms_per_step": round(el_s / 3 * 1e3, 3), "stage_ms_per_step": st_s, "roofline": rf_s,
                            "note": "3 steps after 1 warm-up, stages one after the other; not part of `value`"}
        stage_ms, kern_ms, roof = summarize(prof, args.steps, n_total, full_bytes, ms_per_step, conc, dom_serial)
        if serial_order is not None and conc:
            roof["frac_uncontended"] = serial_order["roofline"]["frac"]
            roof["note"] = ("`achieved` / `frac`: the dominant kernel's time in the TIMED steps, where the coordinate sort and the metrics pass run at the same "
                            "time on streams of their own and take CUs, LDS and bandwidth from it; `frac_uncontended`: the same kernel, same reads, in the "
                            "serial-order steps behind the timed region (serial_order.roofline)")
        if not sfm_mode and args.stages == "full" and order == "metrics":
            roof["concurrent_streams"] = ("the duplication-metrics pass runs under the coordinate sort (side lanes of the context, two host threads): the two "
                                          "stages' kernel times are contended and their sum exceeds what they add to the step; mark duplicates, the BQSR "
                                          "gather and ApplyBQSR run alone; uncontended sort / metrics: `serial_order`")
        what = ("mark duplicates + coordinate sort + optical metrics + BQSR gather + finalize + apply (BASELINE config C3)" if args.stages == "full"
                else "mark duplicates + coordinate sort (BASELINE config C2)")
        out = {
            "metric": "Mreads/s through sort+markdup+BQSR, 150bp PE",
            "value": round(value, 3),
            "unit": "Mreads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": args.scaling if sfm_mode else "weak",
            "vs_baseline": None,
            "dtype": "u8/int32/int64 (integer path; float64 finalize on host)",
            "data": "synthetic",
            "config": {"workload": f"{'C3' if args.stages == 'full' else 'C2'}-style ({mode}): {n_total} reads on rank 0, {args.reads} requested per GPU, 150bp PE, genome {args.genome} "
                                   f"(24 contigs hg38/12), {args.quals} qualities, {what}",
                       "reads_per_gpu": n_total, "max_cycle": MAX_CYCLE,
                       "step_order": (order if not sfm_mode and args.stages == "full" else None),
                       "step_order_probe_ms": order_probe,
                       "resident_rerun": "every timed step starts from the same staged columns (elp_rollback restores FLAG and QUAL); two facts of the "
                                         "staged columns are computed once per staging, not per step: the one-length check of the offset columns "
                                         "(k_uniform_check) and the tile index of the QUAL column (k_flat_index) - together ~0.15 ms that a one-shot "
                                         "`elprep filter` pays once",
                       "parallelism": ("filter: one context" if not sfm_mode else f"sfm: contig groups over {world} GPUs, spread split on one rank, one all-reduce per step "
                                                                              f"({rk.collective} collective)")},
            "roofline": roof,
            "serial_order": serial_order,
            "stage_ms_per_step": stage_ms,
            "kernel_ms_per_step": kern_ms,
            "host_finalize_ms_per_step": round(sum(host_ms[-args.steps:]) / max(len(host_ms[-args.steps:]), 1), 3) if host_ms else None,
            "host_finalize_exposed_ms_per_step": round(sum(wait_ms[-args.steps:]) / max(len(wait_ms[-args.steps:]), 1), 3) if wait_ms else None,
            "staging": {"gen_s": round(gen_s, 2), "h2d_stage_s": round(stage_s, 2),
                        "elp_stage_Mreads_per_s": round(n_total / stage_s / 1e6, 2) if stage_s > 0 else None},
        }
        if os.environ.get("ELP_BENCH_HOST_PARTS") and host_parts:
            out["host_parts_ms"] = {"per_step_total": [round(v, 3) for v in host_ms[-args.steps:]], "per_step_exposed": [round(v, 3) for v in wait_ms[-args.steps:]],
                                    "fetch_finalize_lut_upload": host_parts[-args.steps:]}
        if sfm_mode:
            # what the first multi-GPU record needs to be read without a second run: the collective's share of a step (the wait for the
            # slowest rank included: the call is entered behind a stream sync) and the set-up's record exchange
            ar = rk.allreduce_s[-args.steps:]
            out["allreduce_ms_per_step"] = round(sum(ar) / max(len(ar), 1) * 1e3, 3)
            out["staging"]["route_s"] = round(route_s, 2)
            out["ranks_seen"] = dist.get_world_size()
            out["per_rank_reads"] = per_rank_reads
            out["imbalance_max_over_mean"] = round(max(per_rank_reads) / (sum(per_rank_reads) / len(per_rank_reads)), 4)

    # ---- N = 1 side measurements: config C2 on the same reads, the ~40-quality read set, the PCIe-inclusive staging rate
    if not sfm_mode and rank == 0 and not args.no_extra:
        extra = {}
        try:
            if args.stages == "full":
                for o, wl in (("three", "the coordinate sort, the metrics pass and the BQSR chain (gather, finalize, apply) driven at once behind mark duplicates"),
                              ("metrics", "the metrics pass under the coordinate sort, both behind the gather: the BQSR kernels have the GPU to themselves"),
                              ("serial", "the stages one after the other, as until round 5")):
                    if o == order:
                        continue
                    elo, _ = timed(lambda: step_full(o), restore, 4, 1, eng, barrier)
                    extra["order_" + o] = {"workload": "the main line's step, " + wl, "value": round(n_total / (elo / 4) / 1e6, 3), "unit": "Mreads/s",
                                           "ms_per_step": round(elo / 4 * 1e3, 3)}
        except Exception as e:
            extra["order_ab"] = {"error": repr(e)}
        try:
            if args.stages == "full":
                el, pr = timed(step_c2, restore, 3, 1, eng, barrier)
                st, km, rf = summarize(pr, 3, n_total, BYTES_C2)
                extra["c2_sort_markdup"] = {"workload": f"BASELINE config C2: mark duplicates + coordinate sort of the same {n_total} staged reads",
                                            "value": round(n_total / (el / 3) / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(el / 3 * 1e3, 3),
                                            "stage_ms_per_step": st, "roofline": rf}
        except Exception as e:  # a side measurement must not cost the main line
            extra["c2_sort_markdup"] = {"error": repr(e)}
        eng.close()
        eng = None
        try:
            extra["pcie_inclusive"] = pcie_inclusive(cfg, hdr, min(args.extra_reads, 8_000_000), out["ms_per_step"], n_total)
        except Exception as e:
            extra["pcie_inclusive"] = {"error": repr(e)}
        def side_run(key, workload, mutate, shuffle=False, genome=None, reads=None):
            """the full path on `--extra-reads` reads of a variant of the main workload (data shapes the main line is not tuned on)"""
            try:
                cq = synth.config(genome or args.genome)
                cq.qual_mode = cfg.qual_mode
                mutate(cq)
                hq = cq.header()
                n_want = reads or args.extra_reads
                rs = refs_sites
                if genome and genome != args.genome:  # another genome: its own reference and known sites (made on the generator's thread pool)
                    with ThreadPoolExecutor(workers) as pool:
                        rs = list(pool.map(lambda r: (r, synth.reference(cq, r), flatten_sites(synth.known_sites_raw(cq, r))), range(hq.n_ref)))
                e2 = Engine(hq, dev_id)
                n2 = 0
                rng = np.random.default_rng(7)
                for b in generated([(cq, lo, min(lo + chunk, n_want // 2)) for lo in range(0, n_want // 2, chunk)]):
                    if shuffle:
                        b = b.take(rng.permutation(b.n))
                    e2.stage(b)
                    n2 += b.n
                    del b
                for r, ref, sites in rs:
                    e2.set_reference(r, ref)
                    e2.set_known_sites(r, sites)
                del rs
                e2.sync()
                e2.snapshot()
                sf2, _, rs2 = make_filter_steps(e2, [None])
                side_steps, side_warmup = int(os.environ.get("ELP_BENCH_SIDE_STEPS", "6")), int(os.environ.get("ELP_BENCH_SIDE_WARMUP", "2"))
                el, pr = timed(sf2, rs2, side_steps, side_warmup, e2, barrier)
                st, km, rf = summarize(pr, side_steps, n2, BYTES_FULL_PATH)
                extra[key] = {"workload": f"{n2} reads, {workload}, full path",
                              "value": round(n2 / (el / side_steps) / 1e6, 3), "unit": "Mreads/s", "ms_per_step": round(el / side_steps * 1e3, 3),
                              "steps": side_steps, "warmup": side_warmup,
                              "host_finalize_ms_per_step": round(sum(host_ms[-side_steps:]) / side_steps, 3),
                              "host_finalize_exposed_ms_per_step": round(sum(wait_ms[-side_steps:]) / side_steps, 3),
                              "stage_ms_per_step": st, "kernel_ms_per_step": km, "roofline": rf}
                if os.environ.get("ELP_BENCH_HOST_PARTS"):
                    extra[key]["host_parts_ms"] = {"per_step_total": [round(v, 3) for v in host_ms[-side_steps:]], "per_step_exposed": [round(v, 3) for v in wait_ms[-side_steps:]],
                                                   "fetch_finalize_lut_upload": host_parts[-side_steps:]}
                e2.close()
            except Exception as e:
                extra[key] = {"error": repr(e)}

        if args.extra_reads > 0:
            if args.quals == "binned":
                side_run("full_quals", "same generator with ~40 distinct quality values (3..41 and 2)", lambda c: setattr(c, "qual_mode", 1))
            side_run("shuffled_input", "the main workload's reads staged in random order within every 2 M-record batch (mates are not neighbours: "
                     "the mate table path of mark duplicates)", lambda c: None, shuffle=True)
            side_run("rg16", "the main workload with 16 read groups (16 BQSR covariates instead of 4)", lambda c: setattr(c, "n_lanes", 16))
            side_run("rg32", "the main workload with 32 read groups (32 BQSR covariates)", lambda c: setattr(c, "n_lanes", 32))
        if args.c4_reads > 0:
            # one GPU's share of BASELINE config C4 (30x WGS over 8 GPUs) on the REAL genome shape: hg38's contig lengths (POS needs 28 bits: the
            # coordinate sort runs five radix passes instead of four), 3.1 Gbp of packed reference and known-site flags in HBM
            side_run("c4_share", "genome c4 = hg38 contig lengths (3.1 Gbp; 248 Mbp contigs: 34 live key bits = five radix passes), one GPU's share of a 600 M-read "
                     "30x WGS run over 8 GPUs", lambda c: None, genome="c4", reads=args.c4_reads)
        out["extra"] = extra

    verify_failed = False
    if rank == 0:
        if not sfm_mode and not args.no_cpu_baseline and args.cpu_reads > 0:
            # the CPU leg: timed as the baseline, and its outputs are what the device path is checked against on the same reads
            # (outside every timed region): the bench line carries its own parity proof at the scale the oracle finishes in seconds
            if not sfm_mode and eng is not None:
                eng.close()
                eng = None
            out["cpu_baseline"], ref_out = cpu_baseline(cfg, hdr, args.cpu_reads, refs_sites)
            if not args.no_verify:
                out["verify"] = verify_against_oracle(hdr, refs_sites, ref_out, dev_id)
                verify_failed = not out["verify"]["ok"]
            del ref_out
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if sfm_mode:
        dist.barrier()
        dist.destroy_process_group()
        rk.close()
    elif eng is not None:
        eng.close()
    if verify_failed:
        print("bench.py: device outputs differ from the CPU oracle on the verification sample (see \"verify\" in the line above)", file=sys.stderr)
        sys.exit(3)


def pcie_inclusive(cfg, hdr, n_reads, ms_per_step, n_main):
    """Staging straight from inflated BAM records in page-locked memory (elp_stage_bam: one DMA of the raw bytes, the columns are cut
    on the device) and the way back (elp_emit_sorted_bam: gather in sorted order + D2H), timed: what a host that inflates BGZF blocks
    into pinned buffers gets.  The BAM bytes are written by the generator (tools/synth/bam_writer.c, standing in for the BAM reader)."""
    import ctypes
    from elprep_amd import _lib
    from elprep_amd.engine import Engine
    from tools import synth
    b = synth.generate(cfg, 0, n_reads // 2)
    L = _lib.hip()
    size = synth.bam_records_size(b, hdr.rg_ids)
    ptr = L.elp_pinned_alloc(size)
    buf = np.frombuffer((ctypes.c_uint8 * size).from_address(ptr), dtype=np.uint8)
    _, rec_off = synth.bam_records(b, hdr.rg_ids, out=buf)
    e = Engine(hdr, 0)
    e.set_read_group_ids(hdr.rg_ids)
    e.stage_bam(buf, rec_off=rec_off)  # first call: device allocations (a long-running host reuses its context)
    e.sync()
    e.reset()
    t0 = time.perf_counter()
    e.stage_bam(buf)                   # record starts found by walking the block_size chain on the host
    e.sync()
    t_chain = time.perf_counter() - t0
    e.reset()
    t0 = time.perf_counter()
    e.stage_bam(buf, rec_off=rec_off)  # record starts handed over by the reader
    e.sync()
    t_in = time.perf_counter() - t0
    e.mark_duplicates(True, fetch=False)
    e.sort_coordinate(fetch=False)
    out_ptr = L.elp_pinned_alloc(size + 64)
    out = np.frombuffer((ctypes.c_uint8 * (size + 64)).from_address(out_ptr), dtype=np.uint8)
    t0 = time.perf_counter()
    got = e.emit_sorted_bam(out)
    t_out = time.perf_counter() - t0
    n = b.n
    e.close()
    # the column route: elp_stage_columns from page-locked column buffers (what the cgo binding of INTEGRATION.md uses when the Go host
    # keeps its restaging buffers in C memory), batches of 1 M records as the parse nodes would deliver them
    e = Engine(hdr, 0)
    cols, keep = {}, []
    for name in Engine._STAGE_COLS:
        a = np.ascontiguousarray(getattr(b, name))
        pp = L.elp_pinned_alloc(max(a.nbytes, 8))
        ctypes.memmove(pp, a.ctypes.data, a.nbytes)
        keep.append(pp)
        cols[name] = (pp, a.dtype.itemsize)
    def stage_cols():
        step_n = 1_000_000
        for lo in range(0, n, step_n):
            cnt = min(step_n, n - lo)
            # the per-record columns advance by records, the offset columns too (elp_stage rebases them), the payload columns stay
            ptrs = {k: (pp + lo * isz if k not in ("qname", "cigar", "seq4", "qual") else pp) for k, (pp, isz) in cols.items()}
            e.stage_pointers(cnt, ptrs)
    stage_cols()          # first pass: device allocations
    e.sync()
    e.reset()
    t0 = time.perf_counter()
    stage_cols()
    e.sync()
    t_cols = time.perf_counter() - t0
    col_bytes = sum(np.ascontiguousarray(getattr(b, name)).nbytes for name in Engine._STAGE_COLS)
    for pp in keep:
        L.elp_pinned_free(pp)
    res = {"workload": f"{n} reads as {size} bytes of inflated BAM records in page-locked host memory",
           "stage_columns_pinned_Mreads_per_s": round(n / t_cols / 1e6, 2), "stage_columns_pinned_GB_per_s": round(col_bytes / t_cols / 1e9, 2),
           "stage_bam_Mreads_per_s": round(n / t_in / 1e6, 2), "stage_bam_GB_per_s": round(size / t_in / 1e9, 2),
           "stage_bam_without_record_offsets_Mreads_per_s": round(n / t_chain / 1e6, 2),
           "emit_sorted_bam_Mreads_per_s": round(n / t_out / 1e6, 2), "emit_sorted_bam_GB_per_s": round(got.size / t_out / 1e9, 2),
           # one pass of the path with both transfers, at the main run's step time per read
           "end_to_end_Mreads_per_s": round(n / (t_in + t_out + ms_per_step * 1e-3 * n / max(n_main, 1)) / 1e6, 2)}
    e.close()
    # the BGZF route (round 4): the COMPRESSED blocks cross PCIe, the device inflates them, finds the records and stages them
    # (elp_stage_bgzf); on the way out the device frames the sorted records as BGZF blocks (elp_emit_sorted_bgzf: DEFLATE + CRC-32 on the device).
    # The compressed input is made here with zlib level 1 on a thread pool (it stands for the file on disk)
    try:
        import struct
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        nb_reads = min(n, 4_000_000)
        nbytes = int(rec_off[nb_reads])
        cut = 65280

        def member(k):
            part = bytes(buf[k:min(k + cut, nbytes)])
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            data = co.compress(part) + co.flush()
            return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(data) + 25) + data + struct.pack("<II", zlib.crc32(part), len(part))
        with ThreadPoolExecutor(max(1, min(16, effective_cores()))) as pool:
            bz = np.frombuffer(b"".join(pool.map(member, range(0, nbytes, cut))), dtype=np.uint8)
        e = Engine(hdr, 0)
        e.set_read_group_ids(hdr.rg_ids)
        e.stage_bgzf(bz)  # first call: device allocations
        e.sync()
        e.reset()
        t0 = time.perf_counter()
        e.stage_bgzf(bz)
        e.sync()
        t_bz_in = time.perf_counter() - t0
        # the decoder's kernels by themselves (HIP events around the launches): one launch behind the whole copy, so that a launch's
        # time is not the time it shared the chip with the next chunk's launch
        kern = {}
        try:
            e.reset()
            e.set_tuning("bgzf_copy_chunk", 1 << 30)
            e.profile_enable(True)
            e.profile_reset()
            e.stage_bgzf(bz)
            e.sync()
            prof = e.profile()
            e.profile_enable(False)
            e.set_tuning("bgzf_copy_chunk", 0)
            ms = {k: v[1] for k, v in prof.items()}
            t_inf = (ms.get("stage_bgzf_tokens", 0.0) + ms.get("stage_bgzf_resolve", 0.0)) * 1e-3
            kern = {"stage_bgzf_inflate_kernels_ms": {k: round(v, 3) for k, v in ms.items() if k.startswith("stage_bgzf")},
                    "stage_bgzf_inflate_kernels_GB_per_s": round(nbytes / t_inf / 1e9, 2) if t_inf > 0 else None}
        except Exception as ex:
            kern = {"stage_bgzf_inflate_kernels_error": repr(ex)}
        e.mark_duplicates(True, fetch=False)
        e.sort_coordinate(fetch=False)
        e.emit_sorted_bgzf()
        t0 = time.perf_counter()
        got_bz = e.emit_sorted_bgzf()
        t_bz_out = time.perf_counter() - t0
        e.close()
        res.update({"bgzf_workload": f"{nb_reads} reads: {bz.size} bytes of BGZF blocks (zlib level 1, {bz.size / nbytes:.2f} of the inflated size) from pageable memory in, "
                                     f"{got_bz.size} bytes of BGZF blocks (compressed on the device: DEFLATE with per-block Huffman codes, {got_bz.size / nbytes:.2f} of the inflated size) out",
                    "stage_bgzf_Mreads_per_s": round(nb_reads / t_bz_in / 1e6, 2), "stage_bgzf_inflated_GB_per_s": round(nbytes / t_bz_in / 1e9, 2),
                    "emit_sorted_bgzf_Mreads_per_s": round(nb_reads / t_bz_out / 1e6, 2), "emit_sorted_bgzf_ratio": round(got_bz.size / nbytes, 4),
                    "end_to_end_bgzf_Mreads_per_s": round(nb_reads / (t_bz_in + t_bz_out + ms_per_step * 1e-3 * nb_reads / max(n_main, 1)) / 1e6, 2)})
        res.update(kern)
    except Exception as ex:  # a side measurement must not cost the main line
        res["bgzf_error"] = repr(ex)
    del buf, out
    L.elp_pinned_free(ptr)
    L.elp_pinned_free(out_ptr)
    return res


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


def cpu_baseline(cfg, hdr, n_reads, refs_sites):
    """The CPU oracle (plain-C restatement of the reference's algorithms) on ALL host cores - parallel merge sort with the
    CoordinateLess comparator, sharded duplicate-marking maps, thread-private BQSR tables summed at the end, as the reference's
    pargo-based CPU path is organised - timed on a bounded sample of the same workload.  kind = "port": it is NOT the elPrep binary
    (no Go toolchain and no elprep on the box: profiles/r2a_reference_toolchain_probe.txt).
    -> (the record, the outputs of the run: what verify_against_oracle() compares the device path with)"""
    import oracle as orc
    from tools import synth
    from concurrent.futures import ThreadPoolExecutor
    from elprep_amd.batch import Batch
    cores = effective_cores()
    chunk = 500_000
    with ThreadPoolExecutor(min(cores, 16)) as pool:
        parts = list(pool.map(lambda lo: synth.generate(cfg, lo, min(lo + chunk, n_reads // 2)), range(0, n_reads // 2, chunk)))
    b = Batch.concat(parts) if len(parts) > 1 else parts[0]
    del parts
    refs = [ref for _, ref, _ in refs_sites]
    sites = [st for _, _, st in refs_sites]
    t0 = time.perf_counter()
    flags0 = orc.mark_duplicates_mt(b, hdr, cores)                    # MarkDuplicates while the records stream in (phase 1)
    t_mark = time.perf_counter() - t0
    perm = orc.sort_coordinate_mt(b, flags0, cores)                   # the sort: Finalize of that pipeline
    flags, ctr = orc.dup_metrics_mt(b, hdr, perm, 100, cores)         # MarkOpticalDuplicates over the sorted reads; this entry point marks again
    qt, ct, xt = orc.bqsr_gather_mt(b, hdr, orc.BqsrRef(refs, sites), flags, MAX_CYCLE, cores)
    fin = orc.BqsrFinal(qt, ct, xt, MAX_CYCLE)
    qual = orc.bqsr_apply_mt(fin, b, hdr, 0, (), cores)
    dt_all = time.perf_counter() - t0
    dt = dt_all - t_mark  # the marking inside the metrics call repeats the first one (the reference keeps its maps): counted once
    rec = {"value": round(b.n / dt / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
           "sample": f"{b.n} reads of the same synthetic workload, full path (mark duplicates + sort + optical metrics + BQSR gather + finalize + apply), "
                     f"multithreaded C restatement of the reference algorithms (oracle/, OpenMP, {cores} threads = the CPUs the container is granted: "
                     f"{os.cpu_count()} logical CPUs visible, cgroup quota {cores}), {dt:.1f} s wall = {dt * cores:.0f} core-seconds; the port's metrics "
                     f"entry point repeats the duplicate marking ({t_mark:.1f} s), that repeat is left out of the {dt:.1f} s"}
    return rec, {"batch": b, "flags": flags, "perm": perm, "counters": ctr, "tables": (qt, ct, xt), "qual": qual}


def verify_against_oracle(hdr, refs_sites, ref, dev_id=0):
    """The device path on the reads of the CPU leg, every output against the CPU leg's: duplicate flags, the coordinate-sort
    permutation, the duplication counters, the three BQSR tables, every recalibrated QUAL byte.  Same order of events as step_full.
    Not timed; a mismatch fails the run (exit code 3)."""
    from elprep_amd.engine import BqsrTables, Engine
    b = ref["batch"]
    e = Engine(hdr, dev_id)
    try:
        e.stage(b)
        for r, refseq, sites in refs_sites:
            e.set_reference(r, refseq)
            e.set_known_sites(r, sites)
        flags = e.mark_duplicates(True)
        perm = e.sort_coordinate()
        e.recalibrate_device(MAX_CYCLE)
        qt, ct, xt = e.tables_fetch()
        ctr = e.dup_metrics(100)
        # the LUT as the timed step makes it: the rows form (tables' rows of the counted qualities down, LUT rows + defaults up, the dense LUT
        # expanded on the device)
        quals = e.quals_counted()
        tb = BqsrTables.from_rows(hdr.n_cov, quals, *e.tables_fetch_rows(quals), MAX_CYCLE).finalize()
        e.lut_upload_rows(quals, *tb.build_lut_rows(quals, 0), MAX_CYCLE)
        qual = e.apply_bqsr(None, None, MAX_CYCLE)
    finally:
        e.close()
    oq, oc, ox = ref["tables"]
    res = {"reads": int(b.n), "bases": int(b.qual.size),
           "flags": bool(np.array_equal(flags, ref["flags"])), "perm": bool(np.array_equal(perm, ref["perm"])),
           "counters": bool(np.array_equal(ctr, ref["counters"])),
           "tables": bool(np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)),
           "qual": bool(np.array_equal(qual, ref["qual"])),
           "duplicates": int(((flags & 0x400) != 0).sum()), "table_observations": int(qt[..., 0].sum()),
           "qual_bytes_changed": int((qual != b.qual).sum()),
           "against": "oracle/ (multithreaded C restatement of the reference; parity unpinned, DESIGN.md section 6), same reads, outside the timed region"}
    res["ok"] = all(res[k] for k in ("flags", "perm", "counters", "tables", "qual"))
    return res


if __name__ == "__main__":
    main()
