"""One-process-per-GPU driver for ``update-gtf`` (torch.distributed over RCCL/xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m lr2rmats_amd.dist update-gtf [options] in.bam old.gtf

Every rank parses the inputs with the C host library, takes a contiguous shard of the read array
(balanced by bytes per read; annotation and junction table replicated), runs the gfx950 engine on
its shard, and the per-read result arrays are all-gathered (variable sizes, rank order = read
order).  Rank 0 then runs the order-dependent host tail (split / merge / writers) once, so the
output files are those of the single-GPU run.
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional

import numpy as np

from . import capi, hostlib, workload


def _engine_classify(device_index: int):
    """Default shard classifier: the HIP engine on this rank's GPU.  Returns device views when possible."""
    eng = capi.Engine(device_index)

    def run(job: hostlib.Job, lo: int, hi: int):
        a = job.annotation_arrays()
        r = job.read_arrays()
        eng.set_params(job.prm)
        eng.set_annotation(a["tx_tid"], a["tx_start"], a["tx_end"], a["tx_rev"], a["tx_ex_off"], a["ex_start"], a["ex_end"])
        eng.set_junctions(job.junction_arrays())
        c0, c1 = int(r["cig_off"][lo]), int(r["cig_off"][hi])
        eng.upload_reads(r["tid"][lo:hi], r["pos"][lo:hi], r["rev"][lo:hi], r["cig_off"][lo:hi + 1] - c0, r["cig"][c0:c1], first_read_index=lo)
        eng.run()
        eng.sync()
        return eng.download()
    run.engine = eng
    return run


def run(argv, classify: Optional[Callable] = None, backend: Optional[str] = None) -> int:
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available() and classify is None
    if classify is None and not use_cuda:
        raise RuntimeError("lr2rmats_amd.dist: no GPU visible and no CPU path exists")
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)

    job = hostlib.Job(list(argv), open_outputs=(rank == 0))
    r = job.read_arrays()
    n = int(r["tid"].shape[0])
    weights = 4.0 * np.diff(r["cig_off"]) + 64.0           # ~ bytes a read costs (4c + 21n + 12)
    lo, hi = workload.shard_bounds(n, world, weights)[rank]
    if classify is None:
        classify = _engine_classify(local_rank)
    res = classify(job, lo, hi)

    if world == 1:
        rc = job.finish(res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx)
        job.close()
        return rc

    # all-gatherv of the shard results, rank order = read order
    def gather(arr: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(device)
        outs, _ = workload.all_gatherv(t)
        return [o.cpu().numpy() for o in outs]

    parts = {k: gather(getattr(res, k)) for k in ("ex_start", "ex_end", "ex_flag", "info", "ref_tx")}
    rc = 0
    if rank == 0:
        info = np.concatenate([p.view(np.uint32) for p in parts["info"]])
        ref = np.concatenate([p.view(np.int32) for p in parts["ref_tx"]])
        xs = np.concatenate([p.view(np.int32) for p in parts["ex_start"]])
        xe = np.concatenate([p.view(np.int32) for p in parts["ex_end"]])
        xf = np.concatenate(parts["ex_flag"])
        off = np.zeros(info.shape[0] + 1, np.int64)
        np.cumsum(info >> 8, out=off[1:])
        rc = job.finish(off, xs, xe, xf, info, ref)
    dist.barrier()
    job.close()
    return rc


def main():
    argv = sys.argv[1:]
    if not argv or argv[0] != "update-gtf":
        print("usage: python -m lr2rmats_amd.dist update-gtf [options] <in.bam> <old.gtf>", file=sys.stderr)
        sys.exit(1)
    rc = run(argv)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
