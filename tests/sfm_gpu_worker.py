"""One rank of the two-rank GPU test of the whole `elprep sfm` path (tests/test_gpu_round6.py): a process per rank under
torch.distributed - backend nccl (= RCCL, one GPU per rank: the split phase's records travel by ncclSend / ncclRecv, the tables by the
C ABI's ncclAllReduce) where the box has a GPU per rank, gloo with both ranks on GPU 0 otherwise (the same calls; the records travel
through the device group's send-receive callback, the tables through torch.distributed).
usage: sfm_gpu_worker.py <rank> <world> <port> <out dir> <nccl|gloo>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, backend = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    import torch
    import torch.distributed as dist
    dev_ord = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_ord)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_ord))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from concurrent.futures import ThreadPoolExecutor
        import oracle as orc  # (only its BAM encoder: what a host's BAM reader hands over; the checking is the parent's)
        from elprep_amd import sfm
        from elprep_amd.engine import BqsrTables
        from tests import sfm_worker
        from tools import synth
        cfg, gof, G, owner, b = sfm_worker.make_rank_input(rank, world, pairs_per_rank=2500)
        h = cfg.header()
        comm = sfm.Comm(torch.device("cuda", dev_ord) if backend == "nccl" else torch.device("cpu"))
        rk = sfm.SfmRank(h, dev_ord, comm)
        for e in rk.engines:
            e.set_read_group_ids(h.rg_ids)
        cut = b.n // 2  # two routing rounds: records arrive behind records
        for part in (b.take(np.arange(cut)), b.take(np.arange(cut, b.n))):
            rk.route(part, gof, G, owner, stage=lambda e, x: e.stage_bam(orc.bam_encode(x, h.rg_ids)))
        for r in range(h.n_ref):
            rk.set_reference(r, synth.reference(cfg, r))
            rk.set_known_sites(r, orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))))
        box = {}

        def finalize(qt, ct, xt):
            box["tables"] = (qt.copy(), ct.copy(), xt.copy())
            return BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        with ThreadPoolExecutor(1) as pool:
            ctr = rk.step(500, 100, pool, finalize)
        rk.sync()
        res = {"input": sfm.pack_batch(b), "ctr": ctr, "collective": np.frombuffer(rk.collective.encode(), dtype=np.uint8)}
        for k, name in enumerate(("qt", "ct", "xt")):
            res[name] = box["tables"][k]
        for w, e in enumerate(rk.engines):
            res[f"n{w}"] = np.array([e.n, e.n_sorted], np.int64)
            res[f"flags{w}"] = e.flags() if e.n else np.zeros(0, np.uint16)
            res[f"perm{w}"] = e.permutation()[:e.n_sorted] if e.n else np.zeros(0, np.uint32)
            res[f"qual{w}"] = e.qual() if e.n else np.zeros(0, np.uint8)
        res["merged"] = rk.emit_merged(gof, G, owner)
        np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
        rk.close()
        dist.barrier()
        print("ok", rank, rk.collective, flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
