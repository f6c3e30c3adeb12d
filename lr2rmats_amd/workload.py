"""Bench / multi-GPU helpers: the BASELINE.json workloads, read sharding and the
accepted-record exchange (RCCL all-gatherv through torch.distributed).

Sharding: reads are independent once the cursor values are
computed by binary search, so the coordinate-sorted read array is cut into
contiguous ranges, one per rank; annotation and junction table are replicated.
The only exchange is the all-gatherv of the compacted accepted-novel records
(the input of the order-dependent host merge), in rank order = read order.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import synth

# BASELINE.json "configs"
CONFIGS = {
    "cfg2": dict(n_reads=100_000, n_exons=5, anno_exons=50_000, seed=2, ont=False, micro=0, xs=0.0),
    "cfg3": dict(n_reads=10_000_000, n_exons=8, anno_exons=1_500_000, seed=3, ont=False, micro=0, xs=0.0),
    "cfg5": dict(n_reads=20_000_000, n_exons=12, anno_exons=2_000_000, seed=5, ont=True, micro=3, xs=0.02),
    # cfg3 with 40 isoforms per gene instead of 5 (same exon count, an eighth of the genes): every tile's window holds 33 .. 63 (or, where a tile meets two genes, more)
    # transcripts -- the shape of isoform-rich loci in a real annotation (diagnostics, not a BASELINE config)
    "cfg3_iso40": dict(n_reads=10_000_000, n_exons=8, anno_exons=1_500_000, seed=3, ont=False, micro=0, xs=0.0, tx_per_gene=40),
    # ... and with 100: every window is beyond the 64-bit masks, most exons are shared by transcripts more than 64 apart in file order
    # cfg3 with isoforms per gene drawn heavy-tailed (log-normal, mean about 4.5, up to 200) like a real annotation: most tiles on the
    # 32-bit masks, the isoform-rich loci on the 64-bit-mask and the chunked kernel (bench.py reports it beside the headline)
    "cfg3_gencode": dict(n_reads=10_000_000, n_exons=8, anno_exons=1_500_000, seed=3, ont=False, micro=0, xs=0.0, tx_per_gene="lognormal"),
    "cfg3_iso100": dict(n_reads=10_000_000, n_exons=8, anno_exons=1_500_000, seed=3, ont=False, micro=0, xs=0.0, tx_per_gene=100),
}


def algorithmic_bytes(n_reads: int, n_cigar: int, n_exons: int, n_tx: int, n_anno_exons: int, n_sj: int = 0) -> int:
    """B(r) = 4c + 12 + 8n + (5n - 4) + 8 per read, plus 8E + 20T (+16S) once per launch."""
    return 4 * n_cigar + 12 * n_reads + 8 * n_exons + (5 * n_exons - 4 * n_reads) + 8 * n_reads + \
        8 * n_anno_exons + 20 * n_tx + 16 * n_sj


def shard_bounds(n_reads: int, world: int, weights: np.ndarray | None = None) -> List[Tuple[int, int]]:
    """Contiguous read ranges per rank, balanced by ``weights`` (bytes per read) when given."""
    if weights is None:
        cuts = [(n_reads * k) // world for k in range(world + 1)]
    else:
        cum = np.concatenate([[0], np.cumsum(weights, dtype=np.float64)])
        cuts = [int(np.searchsorted(cum, cum[-1] * k / world)) for k in range(world)] + [n_reads]
        cuts[0] = 0
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def aligned_shard_bounds(tid: np.ndarray, world: int, weights: np.ndarray | None = None):
    """Shard bounds snapped to chromosome boundaries of a tid-sorted read array, or None when the reads are not
    grouped by chromosome.  A rank may get an empty shard when there are fewer chromosomes than ranks."""
    n = int(tid.shape[0])
    if n == 0 or np.any(np.diff(tid) < 0):
        return None
    edges = np.concatenate([[0], np.nonzero(np.diff(tid) != 0)[0] + 1, [n]])       # starts of the chromosomes + n
    ideal = [b[0] for b in shard_bounds(n, world, weights)] + [n]
    cuts = [0]
    for k in range(1, world):
        e = int(edges[np.argmin(np.abs(edges - ideal[k]))])
        cuts.append(max(e, cuts[-1]))
    cuts.append(n)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def make_rank_workload(cfg: dict, rank: int, world: int):
    """Weak-scaling workload: every rank gets ``cfg['n_reads']`` reads of the same mix, drawn from its own
    block of chromosomes, so that the concatenation over ranks is one coordinate-sorted read set of
    ``world * n_reads`` alignments against the one replicated annotation."""
    anno = synth.make_annotation(cfg["anno_exons"], cfg["seed"], mean_tx_exons=cfg["n_exons"] + 1, tx_per_gene=cfg.get("tx_per_gene", 5))
    af = anno.in_file_order()
    nchr = len(anno.chrom_names)
    lo_c, hi_c = (nchr * rank) // world, (nchr * (rank + 1)) // world
    if world == 1:
        sub = anno
    else:
        keep = np.nonzero((anno.tx_tid >= lo_c) & (anno.tx_tid < hi_c))[0]
        lens = np.diff(anno.tx_ex_off)[keep]
        off = np.zeros(len(keep) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        idx = synth._ragged_gather_index(anno.tx_ex_off[keep], lens)
        sub = synth.Annotation(anno.chrom_names, anno.tx_tid[keep], anno.tx_rev[keep], anno.tx_gene[keep], off,
                               anno.ex_start[idx], anno.ex_end[idx], anno.n_genes, None)
    reads = synth.make_reads(sub, cfg["n_reads"], cfg["n_exons"], cfg["seed"] * 1000 + rank, ont=cfg["ont"],
                             micro_exons=cfg["micro"], xs_conflict_frac=cfg["xs"])
    return af, reads


# --------------------------------------------------------------------------- exchange

class _DevArray:
    """Zero-copy torch view of engine-owned HBM through ``__cuda_array_interface__``."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (max(nbytes, 0),), "typestr": "|u1", "data": (ptr or 0, False),
                                         "version": 2, "strides": None}


def device_bytes(ptr: int, nbytes: int, device):
    import torch
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevArray(ptr, nbytes), device=device)


def ragged_gather_index(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Flat indices for concatenating the slices [starts[i], starts[i] + lens[i])."""
    return synth._ragged_gather_index(np.asarray(starts), np.asarray(lens))


def all_gatherv(t, group=None, counts=None):
    """All-gather of 1-D uint8 tensors of different lengths, rank order preserved.

    Counts first (one tiny all-gather), then ONE uniform collective: every rank contributes ``max(counts)`` bytes
    (its payload, padded) to ``all_gather_into_tensor`` -- over RCCL that is a single ring/tree all-gather on xGMI
    instead of a grouped send per rank, and the same code runs on gloo (CPU tests).  ``t`` may be longer than the
    rank's own count (a view of an engine buffer with garbage behind the payload): pass ``counts`` then."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) != "nccl":
        t = t.cpu()             # a CPU transport (gloo: tests that put two ranks on one GPU) -- staged through the host
    if counts is None:
        n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
        allc = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(allc, n, group=group)
        counts = [int(c.item()) for c in allc]
    m = max(counts) if counts else 0
    if m == 0:
        return [t[:0] for _ in range(world)], counts
    if t.numel() >= m:
        mine = t[:m]
    else:
        mine = torch.zeros(m, dtype=t.dtype, device=t.device)
        mine[: t.numel()] = t
    out = torch.empty(world * m, dtype=t.dtype, device=t.device)
    if hasattr(dist, "all_gather_into_tensor") and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
    else:
        dist.all_gather([out[k * m:(k + 1) * m] for k in range(world)], mine.contiguous(), group=group)
    return [out[k * m: k * m + counts[k]] for k in range(world)], counts


def merge_gathered(rec_parts, off_parts, start_parts, end_parts, flag_parts):
    """Concatenate per-rank accepted-record arrays (numpy) in rank order; exon offsets are rebased."""
    rec = np.concatenate(rec_parts)
    lens = [np.diff(o) if len(o) > 1 else np.zeros(0, np.int64) for o in off_parts]
    alll = np.concatenate(lens) if lens else np.zeros(0, np.int64)
    off = np.zeros(len(alll) + 1, np.int64)
    np.cumsum(alll, out=off[1:])
    return rec, off, np.concatenate(start_parts), np.concatenate(end_parts), np.concatenate(flag_parts)
