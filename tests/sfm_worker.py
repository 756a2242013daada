"""Worker of the world_size-2 gloo tests (CPU): one process per rank, rendezvous on 127.0.0.1."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CGS = 80000  # contig-group size for the tiny genome (60k | 45k + 30k): two groups


def make_rank_input(rank, world, pairs_per_rank=1500):
    """what this rank 'reads': the reads of the contig groups it owns, generated group by group (tools/synth home ranges)"""
    from elprep_amd import sfm
    from elprep_amd.batch import Batch
    from tools import synth
    cfg = synth.config("tiny")
    gof, G = sfm.contig_groups(cfg.ref_len, CGS)
    ranges = sfm.group_ranges(gof, G)
    glen = [sum(cfg.ref_len[lo:hi]) for lo, hi in ranges]
    weights = [0.01 * sum(glen)] + glen + [0.03 * sum(glen)]
    owner = sfm.assign_splits(weights, world)
    parts = []
    for g in range(1, G + 1):
        if owner[g] != rank:
            continue
        c = synth.config("tiny")
        c.seed = cfg.seed + 1000 * g
        c.ref_seed = cfg.seed  # one genome: the reference and the known sites do not follow the group's seed
        c.home_lo, c.home_hi = ranges[g - 1]
        npairs = max(1, int(pairs_per_rank * glen[g - 1] / max(sum(glen[k - 1] for k in range(1, G + 1) if owner[k] == rank), 1)))
        parts.append(synth.generate(c, 0, npairs))
    b = Batch.concat(parts) if parts else sfm.empty_batch()
    return cfg, gof, G, owner, b


def worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as orc
        from elprep_amd import sfm
        cfg, gof, G, owner, b = make_rank_input(rank, world)
        comm = sfm.Comm()
        assert comm.world == world and comm.rank == rank
        mine = sfm.route(b, gof, G, owner, comm)
        # tables of this rank's splits with the CPU oracle (the checker; the product computes them on the GPU), then THE all-reduce
        h = cfg.header()
        refs = [__import__("tools.synth", fromlist=["x"]).reference(cfg, r) for r in range(h.n_ref)]
        from tools import synth
        sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
        tot = None
        for part in (mine.local, mine.spread):
            if part.n == 0:
                continue
            perm = orc.sort_coordinate(part)
            flags, ctr, _ = orc.dup_metrics(part, h, perm, 100)
            q, c, x = orc.bqsr_gather(part, h, orc.BqsrRef(refs, sites), flags, 500)
            flat = np.concatenate([q.ravel(), c.ravel(), x.ravel(), ctr.ravel()])
            tot = flat if tot is None else tot + flat
        if tot is None:
            tot = np.zeros(1, np.int64)
        red = comm.allreduce_i64(tot)
        # the send-receive callback the C ABI's device group gets under gloo / on shared GPUs (elp_group_set_p2p <- Comm.sendrecv): one
        # message each way, then one direction only (what elp_exchange_records' header / verdict / payload messages are made of)
        other = 1 - rank
        got = comm.sendrecv(other, bytes([rank + 1]) * (1000 + rank), other, 1000 + other)
        ok = got == bytes([other + 1]) * (1000 + other)
        got = comm.sendrecv(1 if rank == 0 else -1, b"xyz" if rank == 0 else None, 0 if rank == 1 else -1, 3 if rank == 1 else 0)
        ok = ok and ((got is None) if rank == 0 else (got == b"xyz"))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), input=sfm.pack_batch(b), local=sfm.pack_batch(mine.local),
                 spread=sfm.pack_batch(mine.spread), own=tot, reduced=red, sendrecv_ok=np.array([ok]))
    finally:
        dist.destroy_process_group()
