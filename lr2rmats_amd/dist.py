"""One-process-per-GPU driver for ``update-gtf`` (torch.distributed over RCCL/xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m lr2rmats_amd.dist update-gtf [options] in.bam old.gtf

Every rank parses the inputs with the C host library, takes a contiguous shard of the read array
(balanced by bytes per read; annotation and junction table replicated) and runs the gfx950 engine on
its shard.  Then one of two routes gives the output files of the single-GPU run:

* **partitioned** (reads grouped by chromosome, no ``-s`` with a junction table): the shards are cut at
  chromosome boundaries.  The order-dependent tail never looks across chromosomes (``merge_trans`` stops
  at a smaller tid, ``src/update_gtf.c:147``; the novel-exon / site / gene lists likewise), so every rank
  runs split / merge / writers on its own shard in parallel; the only collective is an all-gather of the 16
  summary counters, and rank 0 concatenates the part files in shard order.
* **gathered** (anything else): the per-read result arrays are all-gathered (padded ``all_gather_into_tensor``,
  rank order = read order) and rank 0 runs the tail once.
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

    # The updated GTF of a run without -o belongs on stdout, and nothing else does: libraries print banners there
    # (gloo does, when the process group comes up), so fd 1 is pointed at stderr for the whole run and the GTF is
    # streamed to the saved fd at the end.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

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

    job = hostlib.Job(list(argv), open_outputs=False)
    gtf_tmp = None
    if job.out_path(0) is None:
        import tempfile
        fd, gtf_tmp = tempfile.mkstemp(prefix="l2r_gtf_", dir=os.environ.get("TMPDIR", None))
        os.close(fd)
        if rank != 0:
            os.remove(gtf_tmp)
        if world > 1:                                       # every rank needs the SAME base path for the part files
            obj = [gtf_tmp]
            dist.broadcast_object_list(obj, src=0)
            gtf_tmp = obj[0]
        job.set_out_path(0, gtf_tmp)

    def emit_stdout():
        if gtf_tmp is not None and rank == 0:
            with open(gtf_tmp, "rb") as fh:
                while True:
                    blk = fh.read(1 << 24)
                    if not blk:
                        break
                    os.write(real_stdout, blk)
            os.remove(gtf_tmp)
        os.close(real_stdout)

    r = job.read_arrays()
    n = int(r["tid"].shape[0])
    weights = 4.0 * np.diff(r["cig_off"]) + 64.0           # ~ bytes a read costs (4c + 21n + 12)
    sj = job.junction_arrays()
    aligned = None
    if world > 1 and not (job.prm.split_trans and sj is not None) and os.environ.get("L2R_DIST_GATHER") != "1":
        aligned = workload.aligned_shard_bounds(r["tid"], world, weights)
    bounds = aligned if aligned is not None else workload.shard_bounds(n, world, weights)
    lo, hi = bounds[rank]
    if classify is None:
        classify = _engine_classify(local_rank)
    if hi > lo:
        res = classify(job, lo, hi)
    else:
        z = np.zeros(0, np.int32)
        res = capi.Result(np.zeros(1, np.int64), z, z, np.zeros(0, np.uint8), np.zeros(0, np.uint32), z)

    if world == 1:
        job.open_outputs()
        rc = job.finish(res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx)
        job.close()
        emit_stdout()
        return rc

    if aligned is not None:
        # partitioned route: every rank writes its part, rank 0 concatenates
        suffix = ".part%03d" % rank
        cnt = job.finish_part(lo, hi, res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx, suffix, "", rank == 0)
        t = torch.from_numpy(cnt).to(device)
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allc, t)
        dist.barrier()
        if rank == 0:
            total = sum(c.cpu().numpy() for c in allc)
            for which in range(7):
                path = job.out_path(which)
                if path is None:
                    continue
                with open(path, "wb") as out:
                    for k in range(world):
                        part = path + ".part%03d" % k
                        with open(part, "rb") as fh:
                            while True:
                                blk = fh.read(1 << 24)
                                if not blk:
                                    break
                                out.write(blk)
                        os.remove(part)
            if job.out_path(7):
                job.write_summary(total, job.out_path(7))
        dist.barrier()
        job.close()
        emit_stdout()
        return 0

    if rank == 0:                                           # gathered route: rank 0 owns the output files
        job.open_outputs()
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
    emit_stdout()
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
