"""One-process-per-GPU driver for ``update-gtf`` (torch.distributed over RCCL/xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m lr2rmats_amd.dist update-gtf [options] in.bam old.gtf

Every rank parses the inputs with the C host library, takes a contiguous shard of the read array
(balanced by bytes per read; annotation and junction table replicated) and runs the gfx950 engine on
its shard.  Then one of two routes gives the output files of the single-GPU run:

* **partitioned** (reads coordinate sorted, no ``-s`` with a junction table): the shards are cut at
  chromosome boundaries.  For BAM input the host library makes the cut itself from the records' core fields and every rank
  extracts names, strands and CIGARs of ITS shard only (``hostlib.Job(rank=, world=)``); SAM text is loaded whole and cut here.  The order-dependent tail never looks across chromosomes (``merge_trans`` stops
  at a smaller tid, ``src/update_gtf.c:147``; the novel-exon / site / gene lists likewise), so every rank
  runs split / merge / writers on its own shard in parallel; the only collective is an all-gather of the 16
  summary counters, and rank 0 concatenates the part files in shard order.
* **gathered** (anything else): what the ranks' engines left in HBM is all-gathered from there (padded
  ``all_gather_into_tensor``, no host copy on the way in) and rank 0 runs the tail once.  When no output lists
  every read (``update-gtf ... > new.gtf``, ``-v``, ``-E``: the pipeline's first pass) the message is the engine's
  compacted accepted list (16-byte records + their exons); otherwise it is the per-read result arrays.

Records that are NOT coordinate sorted are classified by rank 0 alone: the annotation / junction cursors of the
reference (``last_anno_i`` / ``last_sj_i``, ``src/update_gtf.c:938``) then depend on every earlier record, which a
shard that starts in the middle does not have (the engine replays that history on the host, one context, one
stream of uploads).  The other ranks idle; the output is the single-process output.
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Optional

import numpy as np

from . import capi, hostlib, workload


def records_sorted(tid: np.ndarray, pos: np.ndarray) -> bool:
    """The engine's criterion (l2r_upload_reads): (tid, pos) never decreases."""
    if tid.shape[0] < 2:
        return True
    t0, t1, p0, p1 = tid[:-1], tid[1:], pos[:-1], pos[1:]
    return not bool(np.any((t1 < t0) | ((t1 == t0) & (p1 < p0))))


class HostShard:
    """Results of a shard in host memory (the CPU classifier the gloo tests plug in)."""

    def __init__(self, res: capi.Result, lo: int, has_sj: bool = False, split: bool = False):
        self.res, self.lo, self.has_sj, self.split = res, lo, has_sj, split

    def _t(self, arr, device):
        import torch
        return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()).to(device)

    def full(self, device):
        r = self.res
        return {k: self._t(getattr(r, k), device) for k in ("info", "ref_tx", "ex_start", "ex_end", "ex_flag")}

    def accepted(self, device):
        r = self.res
        idx = np.nonzero(capi.accepted_mask(r.info, self.has_sj, self.split))[0]
        lens = (r.info[idx] >> 8).astype(np.int64)
        g = workload.ragged_gather_index(r.ex_off[idx], lens)
        rec = np.zeros(idx.shape[0], capi.ACC_REC_DTYPE)
        gi = idx.astype(np.uint64) + np.uint64(self.lo)
        rec["read_lo"] = (gi & np.uint64(0xffffffff)).astype(np.uint32); rec["read_hi"] = (gi >> np.uint64(32)).astype(np.uint32)
        rec["info"] = r.info[idx] | capi.INFO_ACCEPTED; rec["ref_tx"] = r.ref_tx[idx]
        off = np.zeros(idx.shape[0], np.uint32)
        if idx.shape[0] > 1:
            np.cumsum(lens[:-1], out=off[1:])
        return {"rec": self._t(rec, device), "ex_off": self._t(off, device), "ex_start": self._t(r.ex_start[g], device),
                "ex_end": self._t(r.ex_end[g], device), "ex_flag": self._t(r.ex_flag[g], device)}

    def download(self) -> capi.Result:
        return self.res


class DeviceShard:
    """Results of a shard where the engine left them: views of its HBM buffers (zero copy)."""

    def __init__(self, eng: capi.Engine):
        self.eng = eng

    def full(self, device):
        v = self.eng.device_view()
        n, x = int(v.n_reads), int(v.n_exons)
        b = workload.device_bytes
        return {"info": b(v.info, 4 * n, device), "ref_tx": b(v.ref_tx, 4 * n, device), "ex_start": b(v.ex_start, 4 * x, device),
                "ex_end": b(v.ex_end, 4 * x, device), "ex_flag": b(v.ex_flag, x, device)}

    def accepted(self, device):
        v = self.eng.device_view()
        m, x = int(v.n_accepted), int(v.n_accepted_exons)
        b = workload.device_bytes
        return {"rec": b(v.acc_rec, 16 * m, device), "ex_off": b(v.acc_ex_off, 4 * m, device), "ex_start": b(v.acc_ex_start, 4 * x, device),
                "ex_end": b(v.acc_ex_end, 4 * x, device), "ex_flag": b(v.acc_ex_flag, x, device)}

    def download(self) -> capi.Result:
        return self.eng.download()


def _engine_classify(device_index: int):
    """Default shard classifier: the HIP engine on this rank's GPU; the results stay in HBM."""
    eng = capi.Engine(device_index)
    eng.hint_single_run(True)               # (a shard is uploaded and classified once: no tile index, the two-kernel pipeline -- DESIGN.md section 7)

    def run(job: hostlib.Job, lo: int, hi: int, want: int = capi.WANT_RESULTS, base: int = 0):
        a = job.annotation_arrays()
        r = job.read_arrays()
        eng.set_params(job.prm)
        eng.set_outputs(want)
        eng.set_annotation(a["tx_tid"], a["tx_start"], a["tx_end"], a["tx_rev"], a["tx_ex_off"], a["ex_start"], a["ex_end"])
        eng.set_junctions(job.junction_arrays())
        c0, c1 = int(r["cig_off"][lo]), int(r["cig_off"][hi])
        # One engine shard holds fewer than 2^32 reads + CIGAR operations (32-bit exon offsets on the device: host/cmds.c run_engine cuts
        # the one-process CLI's input the same way).  A rank's shard beyond that is classified piece by piece -- consecutive uploads,
        # first_read_index running on, so the engine carries its cursors from piece to piece -- and the results come back to the host
        # (L2R_CHUNK_READS: reads per piece, tests force small pieces with it).
        limit = int(os.environ.get("L2R_CHUNK_READS", "0") or 0)
        max_units = 0xf0000000
        if (hi - lo) + (c1 - c0) < max_units and not (limit and hi - lo > limit):
            eng.upload_reads(r["tid"][lo:hi], r["pos"][lo:hi], r["rev"][lo:hi], r["cig_off"][lo:hi + 1] - c0, r["cig"][c0:c1], first_read_index=base + lo,
                             cig_summary=None if r.get("cig_summary") is None else r["cig_summary"][lo:hi])
            eng.run()
            eng.sync()
            return DeviceShard(eng)
        eng.set_outputs(capi.WANT_RESULTS)                  # (the caller takes what it needs from the full results: HostShard)
        parts, a = [], lo
        # reads + CIGAR operations in front of every record of the shard, computed ONCE (a piece's units = differences of it: the search
        # per piece is a binary search, not two temporaries of the remaining shard's length -- ADVICE r3)
        cum = r["cig_off"][lo:hi + 1].astype(np.int64) + np.arange(hi - lo + 1, dtype=np.int64)
        while a < hi:
            b = lo + max(a - lo + 1, int(np.searchsorted(cum, cum[a - lo] + max_units, side="left")) - 1)
            if limit:
                b = min(b, a + limit)
            b = min(b, hi)
            ca, cb = int(r["cig_off"][a]), int(r["cig_off"][b])
            if (b - a) + (cb - ca) >= 0xfffffff0:
                raise RuntimeError("record %d alone exceeds the engine's shard limit" % (base + a))
            eng.upload_reads(r["tid"][a:b], r["pos"][a:b], r["rev"][a:b], r["cig_off"][a:b + 1] - ca, r["cig"][ca:cb], first_read_index=base + a,
                             cig_summary=None if r.get("cig_summary") is None else r["cig_summary"][a:b])
            eng.run()
            eng.sync()
            parts.append(eng.download())
            a = b
        offs, x = [np.zeros(1, np.int64)], 0
        for q in parts:
            offs.append(q.ex_off[1:] + x)
            x += int(q.ex_off[-1])
        return capi.Result(np.concatenate(offs), np.concatenate([q.ex_start for q in parts]), np.concatenate([q.ex_end for q in parts]),
                           np.concatenate([q.ex_flag for q in parts]), np.concatenate([q.info for q in parts]), np.concatenate([q.ref_tx for q in parts]))
    run.engine = eng
    return run


def assemble_accepted(parts):
    """Rank 0 of the gathered route: the ranks' accepted lists (dicts of numpy byte arrays, rank order) -> one list in
    input order.  A rank's list is made of per-tile chunks in the order its kernels handed them out
    (include/lr2rmats_hip.h, l2r_device_view): the records are sorted by their 64-bit read index and the exons laid
    out record by record.  Returns (read_idx, capi.Result of the accepted rows)."""
    recs, offs, base = [], [], 0
    for p in parts:
        rec = p["rec"].view(capi.ACC_REC_DTYPE)
        recs.append(rec)
        offs.append(p["ex_off"].view(np.uint32).astype(np.int64) + base)
        base += p["ex_start"].view(np.int32).shape[0]
    rec = np.concatenate(recs) if recs else np.zeros(0, capi.ACC_REC_DTYPE)
    off = np.concatenate(offs) if offs else np.zeros(0, np.int64)
    xs = np.concatenate([p["ex_start"].view(np.int32) for p in parts]) if parts else np.zeros(0, np.int32)
    xe = np.concatenate([p["ex_end"].view(np.int32) for p in parts]) if parts else np.zeros(0, np.int32)
    xf = np.concatenate([p["ex_flag"].view(np.uint8) for p in parts]) if parts else np.zeros(0, np.uint8)
    idx = rec["read_lo"].astype(np.int64) | (rec["read_hi"].astype(np.int64) << 32)
    order = np.argsort(idx, kind="stable")
    rec, off, idx = rec[order], off[order], idx[order]
    lens = (rec["info"] >> 8).astype(np.int64)
    g = workload.ragged_gather_index(off, lens)
    new_off = np.zeros(rec.shape[0] + 1, np.int64)
    np.cumsum(lens, out=new_off[1:])
    return idx, capi.Result(new_off, xs[g], xe[g], xf[g], rec["info"].copy(), rec["ref_tx"].copy())


def run(argv, classify: Optional[Callable] = None, backend: Optional[str] = None) -> int:
    """One rank of `update-gtf` (see the module's head).  A process group that this call brings up goes down with it: a rank that
    leaves the interpreter with gloo's threads still running can abort in their teardown ("terminate called without an active
    exception"), which turned a finished run into exit code -6 now and then."""
    import torch.distributed as dist
    had_group = dist.is_available() and dist.is_initialized()
    try:
        return _run(argv, classify, backend)
    finally:
        if not had_group and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


def _run(argv, classify: Optional[Callable] = None, backend: Optional[str] = None) -> int:
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
        # (L2R_DIST_BACKEND=gloo: tests that put several ranks on one GPU, which RCCL does not allow)
        dist.init_process_group(backend or os.environ.get("L2R_DIST_BACKEND") or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)

    # (a rank of a multi-process run loads only its shard of a coordinate-sorted BAM: hostlib.Job.shard)
    job = hostlib.Job(list(argv), open_outputs=False, rank=rank, world=world)
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

    sharded, shard_lo, shard_hi, n_total = job.shard()
    inflated = (0, 0)
    if world > 1:
        # A rank that has inflated only the BGZF blocks of its own records (shard() == 2: host/aln_reader.c h_read_alignments_blocks) found
        # its start without the records in front of it: the ranks compare notes.  Rank r's start must be rank r - 1's end (which that
        # rank reached record by record), every rank's records must be coordinate sorted, and so must the ranks one after the other.
        # If anything is off -- or some rank could not read the file that way -- those who did load it the other way (the whole file
        # inflated, cut by the records' weights), like the others.
        mine = {"mode": sharded}
        if sharded == 2:
            info = job.shard_blocks()
            rr = job.read_arrays()
            mine.update({"start": tuple(info[0:2]), "end": tuple(info[2:4]), "n": int(rr["tid"].shape[0]), "bytes": (info[4], info[5]),
                         "sorted": records_sorted(rr["tid"], rr["pos"]),
                         "first": (int(np.uint32(rr["tid"][0])), int(rr["pos"][0])) if rr["tid"].shape[0] else None,
                         "last": (int(np.uint32(rr["tid"][-1])), int(rr["pos"][-1])) if rr["tid"].shape[0] else None})
        every = [None] * world
        dist.all_gather_object(every, mine)
        if any(e["mode"] == 2 for e in every):
            ok = all(e["mode"] == 2 for e in every)
            ok = ok and all(e["sorted"] for e in every) and all(every[k]["end"] == every[k + 1]["start"] for k in range(world - 1))
            keys = [e for e in every if ok and e["n"]]
            ok = ok and all(keys[k]["last"] <= keys[k + 1]["first"] for k in range(len(keys) - 1))
            if ok:
                sharded = 1
                shard_lo = sum(e["n"] for e in every[:rank]); shard_hi = shard_lo + mine["n"]; n_total = sum(e["n"] for e in every)
                inflated = mine["bytes"]
            else:
                if rank == 0:
                    print("[lr2rmats_amd.dist] the ranks' block ranges do not meet: every rank reads the whole file", file=sys.stderr)
                os.environ["L2R_DIST_BLOCKS"] = "0"
                if sharded == 2:
                    job.close()
                    job = hostlib.Job(list(argv), open_outputs=False, rank=rank, world=world)
                    if gtf_tmp is not None:
                        job.set_out_path(0, gtf_tmp)
                    sharded, shard_lo, shard_hi, n_total = job.shard()
    r = job.read_arrays()
    n = int(r["tid"].shape[0])
    weights = 4.0 * np.diff(r["cig_off"]) + 64.0           # ~ bytes a read costs (4c + 21n + 12)
    sj = job.junction_arrays()
    aligned = None
    in_order = True if sharded else records_sorted(r["tid"], r["pos"])
    if world > 1 and not sharded and in_order and not (job.prm.split_trans and sj is not None) and os.environ.get("L2R_DIST_GATHER") != "1":
        aligned = workload.aligned_shard_bounds(r["tid"], world, weights)
    if os.environ.get("L2R_DIST_SHARD_TRACE"):                # tests: what this rank loaded
        with open("%s.%d" % (os.environ["L2R_DIST_SHARD_TRACE"], rank), "w") as fh:
            fh.write("%d %d %d %d %d %d\n" % (1 if sharded else 0, shard_lo, shard_hi, n_total, inflated[0], inflated[1]))      # (+ compressed bytes inflated / in the file: block ranges only)
    base = 0
    if sharded:
        # the host library has cut the records already (same rule, chromosome-aligned): this rank holds [shard_lo, shard_hi) only
        aligned = [(0, n)] * world
        base = shard_lo
    if aligned is not None:
        bounds = aligned
    elif world > 1 and not in_order:
        bounds = [(0, n)] + [(n, n)] * (world - 1)         # cursor history: one rank, one stream of records (see the module text)
        if rank == 0:
            print("[lr2rmats_amd.dist] records are not coordinate sorted: rank 0 classifies all of them", file=sys.stderr)
    else:
        bounds = workload.shard_bounds(n, world, weights)
    lo, hi = bounds[rank]
    accepted_only = aligned is None and world > 1 and not job.needs_all_reads()
    if rank == 0 and os.environ.get("L2R_DIST_TRACE"):      # tests: which route ran
        with open(os.environ["L2R_DIST_TRACE"], "w") as fh:
            fh.write(("partitioned" if aligned is not None else ("one rank, " if (world > 1 and not in_order) else "") + "gathered " +
                      ("accepted" if accepted_only else "full")) + "\n")
    if classify is None:
        classify = _engine_classify(local_rank)
    z = np.zeros(0, np.int32)
    empty = HostShard(capi.Result(np.zeros(1, np.int64), z, z, np.zeros(0, np.uint8), np.zeros(0, np.uint32), z), lo)
    if hi > lo:
        # (the engine is told which output to make; a plugged-in classifier returns a capi.Result of its shard)
        if hasattr(classify, "engine"):
            shard = classify(job, lo, hi, capi.WANT_ACCEPTED if accepted_only else capi.WANT_RESULTS, base)
        else:
            shard = classify(job, lo, hi)
        if isinstance(shard, capi.Result):
            shard = HostShard(shard, lo, sj is not None, bool(job.prm.split_trans))
    else:
        shard = empty

    if world == 1:
        res = shard.download()
        job.open_outputs()
        rc = job.finish(res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx)
        job.close()
        emit_stdout()
        return rc

    if aligned is not None:
        # partitioned route: every rank writes its part, rank 0 concatenates
        suffix = ".part%03d" % rank
        res = shard.download()
        cnt = job.finish_part(lo, hi, res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx, suffix, "", rank == 0)
        # gene lists: an id equal to the last entry of the parts before this one is not counted again (hostlib.Job.part_last_genes)
        lasts = [None] * world
        dist.all_gather_object(lasts, job.part_last_genes())
        for q in range(2):
            before = [l[q] for l in lasts[:rank] if l[q] is not None]
            if before and job.part_has_first_gene(q, before[-1]):
                cnt[job.GENE_COUNTERS[q]] -= 1
        t = torch.from_numpy(cnt).to(device if dist.get_backend() == "nccl" else "cpu")
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
    # all-gatherv of what the shards left on the device, rank order = read order; one host copy on rank 0
    def gather(tensors: dict):
        out = []
        for k in sorted(tensors):
            parts, _ = workload.all_gatherv(tensors[k])
            out.append((k, [p.cpu().numpy() for p in parts] if rank == 0 else None))
        return [{k: parts[w] for k, parts in out} for w in range(world)] if rank == 0 else None

    rc = 0
    if accepted_only:
        parts = gather(shard.accepted(device))
        if rank == 0:
            idx, res = assemble_accepted(parts)
            rc = job.finish_accepted(idx, res.ex_off, res.ex_start, res.ex_end, res.ex_flag, res.info, res.ref_tx)
    else:
        parts = gather(shard.full(device))
        if rank == 0:
            info = np.concatenate([p["info"].view(np.uint32) for p in parts])
            ref = np.concatenate([p["ref_tx"].view(np.int32) for p in parts])
            xs = np.concatenate([p["ex_start"].view(np.int32) for p in parts])
            xe = np.concatenate([p["ex_end"].view(np.int32) for p in parts])
            xf = np.concatenate([p["ex_flag"].view(np.uint8) for p in parts])
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
