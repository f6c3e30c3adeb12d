"""Seeded synthetic inputs for the lr2rmats update-gtf path.

Produces, deterministically from a seed,

* an annotation (``Annotation``): genes on 24 chromosomes, 5 transcripts per
  gene drawn from a 16-exon pool, as structure-of-arrays plus the GTF text;
* coordinate-sorted long-read alignments (``Reads``) as structure-of-arrays
  (``tid, pos, rev, cig_off, cig`` -- exactly what a BAM record carries for this
  path, cf. reference ``src/bam2gtf.c:31-37``) plus the SAM text;
* an optional STAR ``SJ.out.tab`` table.

The read mix follows SURVEY.md section 8(d): exact sub-chains, ragged ends,
exon skips, alternative donors, shifted (unrecognised) and single-exon reads,
with a full-length fraction, and an ONT-like mode (indel-riddled CIGARs,
micro-exons, XS tags contradicting FLAG) for config 5.

Nothing here is on the product path: it feeds tests and bench.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

CIG_M, CIG_I, CIG_D, CIG_N, CIG_S = 0, 1, 2, 3, 4
_CIG_CHARS = "MIDNSHP=XB"


# --------------------------------------------------------------------------- annotation

@dataclass
class Annotation:
    chrom_names: List[str]
    tx_tid: np.ndarray       # int32 [T]   (BAM header order)
    tx_rev: np.ndarray       # uint8 [T]
    tx_gene: np.ndarray      # int32 [T]   gene index
    tx_ex_off: np.ndarray    # int64 [T+1]
    ex_start: np.ndarray     # int32 [E]   ascending inside a transcript, 1-based closed
    ex_end: np.ndarray       # int32 [E]
    n_genes: int
    order: Optional[np.ndarray] = None   # file order of transcripts (indices into the arrays)

    @property
    def n_tx(self) -> int:
        return int(self.tx_tid.shape[0])

    @property
    def n_exons(self) -> int:
        return int(self.ex_start.shape[0])

    @property
    def tx_start(self) -> np.ndarray:
        return self.ex_start[self.tx_ex_off[:-1]]

    @property
    def tx_end(self) -> np.ndarray:
        return self.ex_end[self.tx_ex_off[1:] - 1]

    def gene_id(self, g: int) -> str:
        return "SYNG%08d" % g

    def gene_name(self, g: int) -> str:
        return "sg%d" % g

    def tx_id(self, t: int) -> str:
        return "SYNT%08d" % t

    def tx_name(self, t: int) -> str:
        return "sg%d-%d" % (int(self.tx_gene[t]), t)

    def file_order(self) -> np.ndarray:
        return self.order if self.order is not None else np.arange(self.n_tx)

    def in_file_order(self) -> "Annotation":
        """The same annotation with transcripts re-indexed in GTF file order (what a reader sees)."""
        o = self.file_order()
        if self.order is None:
            return self
        lens = np.diff(self.tx_ex_off)[o]
        off = np.zeros(len(o) + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        idx = _ragged_gather_index(self.tx_ex_off[o], lens)
        return Annotation(self.chrom_names, self.tx_tid[o], self.tx_rev[o], self.tx_gene[o], off,
                          self.ex_start[idx], self.ex_end[idx], self.n_genes, None)

    def write_gtf(self, path: str, with_transcript_rows: bool = True, source: str = "synth") -> None:
        with open(path, "w") as fh:
            fh.write("#!synthetic annotation\n")
            for t in self.file_order():
                t = int(t)
                chrom = self.chrom_names[int(self.tx_tid[t])]
                strand = "-" if self.tx_rev[t] else "+"
                g = int(self.tx_gene[t])
                attr = 'gene_id "%s"; transcript_id "%s"; gene_name "%s"; transcript_name "%s";' % (
                    self.gene_id(g), self.tx_id(t), self.gene_name(g), self.tx_name(t))
                a, b = int(self.tx_ex_off[t]), int(self.tx_ex_off[t + 1])
                if with_transcript_rows:
                    fh.write("%s\t%s\ttranscript\t%d\t%d\t.\t%s\t.\t%s\n" % (
                        chrom, source, self.ex_start[a], self.ex_end[b - 1], strand, attr))
                rng = range(b - 1, a - 1, -1) if self.tx_rev[t] else range(a, b)
                for k in rng:
                    fh.write("%s\t%s\texon\t%d\t%d\t.\t%s\t.\t%s\n" % (
                        chrom, source, self.ex_start[k], self.ex_end[k], strand, attr))


def _ragged_gather_index(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Flat indices for concatenating slices [starts[i], starts[i]+lens[i])."""
    lens = lens.astype(np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    out_off = np.zeros(len(lens), np.int64)
    np.cumsum(lens[:-1], out=out_off[1:])
    return np.arange(total, dtype=np.int64) - np.repeat(out_off, lens) + np.repeat(starts.astype(np.int64), lens)


def _segment_reduce(val: np.ndarray, off: np.ndarray, ufunc, identity: int) -> np.ndarray:
    """ufunc-reduction of val[off[i]:off[i+1]] for every i (identity for empty segments)."""
    n = len(off) - 1
    out = np.full(n, identity, np.int64)
    if n == 0 or len(val) == 0:
        return out
    full = off[1:] > off[:-1]
    if full.any():
        v = np.concatenate([val.astype(np.int64), [identity]])           # (a sentinel: trailing empty segments index it)
        red = ufunc.reduceat(v, off[:-1].astype(np.int64))
        out[full] = red[full]
    return out


def cigar_summary(cig_off: np.ndarray, cig: np.ndarray) -> np.ndarray:
    """What a reader knows of every record's CIGAR while it converts it (include/lr2rmats_hip.h, l2r_reads::cig_summary): [N, 3] uint32 --
    reference bases | N operations + the shortest of them << 16 | the longest D operation + the shortest stretch of reference bases
    between two N operations << 16 (each of the four saturated at 65535; 65535 where there is no N / no such stretch).
    host/aln_reader.c makes the same words for SAM / BAM input; this is the generator's (and the tests') form of it."""
    off = np.asarray(cig_off, np.int64)
    c = np.asarray(cig, np.uint32)
    n = len(off) - 1
    op = (c & 15).astype(np.int64)
    ln = (c >> 4).astype(np.int64)
    adv = np.isin(op, (0, 2, 3, 7, 8))
    ref = np.where(adv, ln, 0)
    cs = np.concatenate([[0], np.cumsum(ref)])
    ref_len = cs[off[1:]] - cs[off[:-1]]
    is_n = op == 3
    cn = np.concatenate([[0], np.cumsum(is_n)])
    n_n = cn[off[1:]] - cn[off[:-1]]
    sat = 65535
    min_n = _segment_reduce(np.where(is_n, np.minimum(ln, sat), sat), off, np.minimum, sat)
    max_d = _segment_reduce(np.where(op == 2, np.minimum(ln, sat), 0), off, np.maximum, 0)
    # stretches between two N operations of one record: reference bases of the other ops between them
    cx = np.concatenate([[0], np.cumsum(np.where(is_n, 0, ref))])       # cx[k] = such bases in front of op k (all records, running)
    idx = np.nonzero(is_n)[0]
    seg = np.full(len(idx), sat, np.int64)
    if len(idx) > 1:
        rid = np.searchsorted(off, idx, side="right") - 1                # the record of every N operation
        same = rid[1:] == rid[:-1]
        seg[1:] = np.where(same, np.minimum(cx[idx[1:]] - cx[idx[:-1]], sat), sat)
    min_seg = _segment_reduce(seg, cn[off], np.minimum, sat)
    out = np.empty((n, 3), np.uint32)
    out[:, 0] = np.minimum(ref_len, 0xffffffff).astype(np.uint32)
    out[:, 1] = (np.minimum(n_n, sat) | (min_n << 16)).astype(np.uint32)
    out[:, 2] = (max_d | (min_seg << 16)).astype(np.uint32)
    return out


def make_annotation(n_exons: int, seed: int, nchr: int = 24, tx_per_gene=5, pool: int = 16,
                    mean_tx_exons: int = 10, shuffle_within_gene: bool = False,
                    long_tx_per_chrom: int = 0, single_exon_tx_frac: float = 0.03) -> Annotation:
    """About ``n_exons`` exon rows: genes = n_exons / (tx_per_gene * mean_tx_exons).
    ``tx_per_gene="lognormal"``: isoforms per gene drawn heavy-tailed like a real annotation's (log-normal, mean about 4.5,
    sigma 1.1, at most 200: about 1 % of the genes beyond 32 isoforms, 0.15 % beyond 64) -- reads sample transcripts
    uniformly, so isoform-rich genes get reads in proportion."""
    rng = np.random.default_rng([seed, 0xA770])
    per_gene = None
    if isinstance(tx_per_gene, str):
        if tx_per_gene != "lognormal":
            raise ValueError("tx_per_gene: an integer or 'lognormal'")
        n_genes = max(1, int(n_exons / (4.5 * mean_tx_exons)))
        per_gene = np.clip(np.rint(rng.lognormal(np.log(4.5) - 0.605, 1.1, size=n_genes)), 1, 200).astype(np.int64)
    else:
        n_genes = max(1, n_exons // (tx_per_gene * mean_tx_exons))
    per_chr = -(-n_genes // nchr)
    chrom_names = ["chr%d" % (i + 1) for i in range(nchr)]
    # pool exons per gene
    ex_len = rng.integers(50, 301, size=(n_genes, pool))
    in_len = rng.integers(100, 3001, size=(n_genes, pool))
    gene_chr = (np.arange(n_genes) // per_chr).astype(np.int32)
    gene_slot = np.arange(n_genes) % per_chr
    gene_base = 10_000 + gene_slot * 60_000 + rng.integers(0, 5001, size=n_genes)
    rel_start = np.cumsum(ex_len + in_len, axis=1) - (ex_len + in_len)      # start of exon k relative to base
    p_start = (gene_base[:, None] + rel_start).astype(np.int64)
    p_end = p_start + ex_len - 1
    gene_rev = rng.integers(0, 2, size=n_genes).astype(np.uint8)

    if per_gene is None:
        per_gene = np.full(n_genes, int(tx_per_gene), np.int64)
    n_tx = int(per_gene.sum())
    tx_gene = np.repeat(np.arange(n_genes, dtype=np.int32), per_gene)
    want = np.clip(mean_tx_exons + rng.integers(-3, 4, size=n_tx), 2, pool)
    single = rng.random(n_tx) < single_exon_tx_frac
    want[single] = 1
    # choose `want` exons out of the pool: rank random keys
    keys = rng.random((n_tx, pool))
    ranks = np.argsort(np.argsort(keys, axis=1), axis=1)
    pick = ranks < want[:, None]                                   # [T, pool] bool, in pool (= genomic) order
    tx_ex_off = np.zeros(n_tx + 1, np.int64)
    np.cumsum(pick.sum(axis=1), out=tx_ex_off[1:])
    ex_start = p_start[tx_gene][pick].astype(np.int32)
    ex_end = p_end[tx_gene][pick].astype(np.int32)
    tx_tid = gene_chr[tx_gene]
    tx_rev = gene_rev[tx_gene]

    if long_tx_per_chrom > 0:
        # adversarial: multi-Mb two-exon transcripts early on each chromosome that pin the
        # reference's annotation cursor (SURVEY.md section 7 "hard parts")
        extra_tid, extra_s, extra_e = [], [], []
        for c in range(nchr):
            for k in range(long_tx_per_chrom):
                s0 = 5_000 + 37 * k
                e1 = int(gene_base[gene_chr == c].max(initial=100_000)) + 40_000 - 11 * k
                extra_tid.append(c)
                extra_s.append([s0, e1 - 200])
                extra_e.append([s0 + 150, e1])
        m = len(extra_tid)
        tx_tid = np.concatenate([tx_tid, np.array(extra_tid, np.int32)])
        tx_rev = np.concatenate([tx_rev, np.zeros(m, np.uint8)])
        tx_gene = np.concatenate([tx_gene, np.arange(n_genes, n_genes + m, dtype=np.int32)])
        ex_start = np.concatenate([ex_start, np.array(extra_s, np.int32).ravel()])
        ex_end = np.concatenate([ex_end, np.array(extra_e, np.int32).ravel()])
        tx_ex_off = np.concatenate([tx_ex_off, tx_ex_off[-1] + 2 * np.arange(1, m + 1, dtype=np.int64)])
        n_genes += m
        n_tx += m

    tstart = ex_start[tx_ex_off[:-1]]
    tend = ex_end[tx_ex_off[1:] - 1]
    if shuffle_within_gene:
        # GENCODE style: genes sorted, transcripts inside a gene in arbitrary order
        gstart = np.full(n_genes, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(gstart, tx_gene, tstart.astype(np.int64))
        order = np.lexsort((rng.random(n_tx), gstart[tx_gene], tx_tid))
    else:
        order = np.lexsort((tend, tstart, tx_tid))
    return Annotation(chrom_names, tx_tid.astype(np.int32), tx_rev.astype(np.uint8), tx_gene, tx_ex_off,
                      ex_start, ex_end, n_genes, order.astype(np.int64))


# --------------------------------------------------------------------------- reads

@dataclass
class Reads:
    chrom_names: List[str]
    tid: np.ndarray        # int32 [N]
    pos: np.ndarray        # int32 [N]  0-based leftmost
    rev: np.ndarray        # uint8 [N]  strand as the path sees it (XS:A wins over FLAG & 16)
    flag_rev: np.ndarray   # uint8 [N]  FLAG & 16 as written to SAM
    has_xs: np.ndarray     # uint8 [N]
    cig_off: np.ndarray    # int64 [N+1]
    cig: np.ndarray        # uint32 [sum]  len << 4 | op
    name_base: int = 0
    chrom_len: int = 250_000_000
    sorted: bool = True
    extra: dict = field(default_factory=dict)

    @property
    def n(self) -> int:
        return int(self.tid.shape[0])

    def qname(self, i: int) -> str:
        return "read%08d" % (self.name_base + i)

    def slice(self, a: int, b: int) -> "Reads":
        lo, hi = int(self.cig_off[a]), int(self.cig_off[b])
        return Reads(self.chrom_names, self.tid[a:b], self.pos[a:b], self.rev[a:b], self.flag_rev[a:b],
                     self.has_xs[a:b], self.cig_off[a:b + 1] - lo, self.cig[lo:hi], self.name_base + a,
                     self.chrom_len, self.sorted)

    def cigar_string(self, i: int) -> str:
        ops = self.cig[int(self.cig_off[i]):int(self.cig_off[i + 1])]
        if len(ops) == 0:
            return "*"
        return "".join("%d%s" % (int(c) >> 4, _CIG_CHARS[int(c) & 15]) for c in ops)

    def write_sam(self, path: str, sort_order: Optional[str] = None) -> None:
        so = sort_order or ("coordinate" if self.sorted else "unsorted")
        with open(path, "w") as fh:
            fh.write("@HD\tVN:1.6\tSO:%s\n" % so)
            for c in self.chrom_names:
                fh.write("@SQ\tSN:%s\tLN:%d\n" % (c, self.chrom_len))
            fh.write("@PG\tID:synth\tPN:lr2rmats_amd.synth\n")
            for i in range(self.n):
                flag = 16 if self.flag_rev[i] else 0
                aux = ""
                if self.has_xs[i]:
                    aux = "\tXS:A:%s" % ("-" if self.rev[i] else "+")
                fh.write("%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*%s\n" % (
                    self.qname(i), flag, self.chrom_names[int(self.tid[i])], int(self.pos[i]) + 1,
                    self.cigar_string(i), aux))


def write_bam(reads: "Reads", path: str, level: int = 1, long_cigar_as_cg: bool = True) -> None:
    """Minimal BAM (BGZF) writer for tests: header with @SQ lines, one record per read, SEQ/QUAL absent
    (l_seq = 0), XS:A aux when ``has_xs``; CIGARs of more than 65535 ops go to the CG:B,I tag as SAMv1 4.2.2 says."""
    import struct
    import zlib
    text = "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if reads.sorted else "unsorted")
    text += "".join("@SQ\tSN:%s\tLN:%d\n" % (c, reads.chrom_len) for c in reads.chrom_names)
    raw = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(reads.chrom_names)))
    for c in reads.chrom_names:
        raw += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", reads.chrom_len)
    for i in range(reads.n):
        name = reads.qname(i).encode() + b"\0"
        ops = reads.cig[int(reads.cig_off[i]):int(reads.cig_off[i + 1])].astype("<u4")
        aux = b""
        if reads.has_xs[i]:
            aux += b"XSA" + (b"-" if reads.rev[i] else b"+")
        cig_bytes = ops.tobytes()
        n_cig = len(ops)
        if n_cig > 65535 and long_cigar_as_cg:
            reflen = int(sum(int(c) >> 4 for c in ops if (int(c) & 15) in (0, 2, 3, 7, 8)))
            aux += b"CGBI" + struct.pack("<I", n_cig) + cig_bytes
            cig_bytes = struct.pack("<II", (0 << 4) | 4, (reflen << 4) | 3)
            n_cig = 2
        flag = 16 if reads.flag_rev[i] else 0
        body = struct.pack("<iiBBHHHIiii", int(reads.tid[i]), int(reads.pos[i]), len(name), 60, 4680, n_cig, flag, 0, -1, -1, 0)
        body += name + cig_bytes + aux
        raw += struct.pack("<I", len(body)) + body
    with open(path, "wb") as fh:
        def block(data: bytes):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(data) + co.flush()
            bsize = len(comp) + 25
            fh.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize))
            fh.write(comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
        for a in range(0, len(raw), 0xff00):
            block(bytes(raw[a:a + 0xff00]))
        block(b"")


def write_bam_fast(reads: "Reads", path: str, level: int = 1, threads: Optional[int] = None, chunk: int = 400_000) -> None:
    """The same file format as ``write_bam`` (BGZF blocks of 0xff00 raw bytes, l_seq = 0 records), built with numpy
    scatter writes and compressed on several threads: 10 M reads in seconds instead of minutes (bench.py's end-to-end
    leg writes its input with this).  CIGARs beyond 65535 ops are not supported here (use ``write_bam``)."""
    import os
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    text = "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if reads.sorted else "unsorted")
    text += "".join("@SQ\tSN:%s\tLN:%d\n" % (c, reads.chrom_len) for c in reads.chrom_names)
    head = bytearray(b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(reads.chrom_names)))
    for c in reads.chrom_names:
        head += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", reads.chrom_len)
    fixed_dt = np.dtype([("bs", "<u4"), ("tid", "<i4"), ("pos", "<i4"), ("lrn", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"),
                         ("flag", "<u2"), ("lseq", "<u4"), ("ntid", "<i4"), ("npos", "<i4"), ("tlen", "<i4"), ("name", "u1", (13,))])
    assert fixed_dt.itemsize == 49
    BLOCK = 0xff00

    def block(data: bytes) -> bytes:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(data) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))

    nthr = threads or min(32, os.cpu_count() or 1)
    pending = bytes(head)
    with open(path, "wb") as fh, ThreadPoolExecutor(nthr) as pool:
        for a in range(0, max(reads.n, 1), chunk):
            b = min(reads.n, a + chunk)
            n = b - a
            if n > 0:
                c0 = int(reads.cig_off[a])
                nc = np.diff(reads.cig_off[a:b + 1]).astype(np.int64)
                if nc.max(initial=0) > 65535:
                    raise ValueError("write_bam_fast: CIGAR with more than 65535 ops (use write_bam)")
                xs = reads.has_xs[a:b].astype(np.int64)
                body = 45 + 4 * nc + 4 * xs
                off = np.zeros(n + 1, np.int64)
                np.cumsum(body + 4, out=off[1:])
                buf = np.zeros(int(off[-1]), np.uint8)
                fx = np.zeros(n, fixed_dt)
                fx["bs"] = body; fx["tid"] = reads.tid[a:b]; fx["pos"] = reads.pos[a:b]; fx["lrn"] = 13; fx["mapq"] = 60; fx["bin"] = 4680
                fx["ncig"] = nc; fx["flag"] = np.where(reads.flag_rev[a:b] != 0, 16, 0); fx["ntid"] = -1; fx["npos"] = -1
                idx = reads.name_base + np.arange(a, b, dtype=np.int64)
                nm = fx["name"]
                nm[:, 0:4] = np.frombuffer(b"read", np.uint8)
                for k in range(8):
                    nm[:, 4 + k] = (idx // 10 ** (7 - k)) % 10 + 48
                rows = fx.view(np.uint8).reshape(n, 49)
                buf[(off[:n, None] + np.arange(49)[None, :]).ravel()] = rows.ravel()
                m = int(nc.sum())
                if m:
                    words = np.ascontiguousarray(reads.cig[c0:c0 + m]).astype("<u4").view(np.uint8).reshape(m, 4)
                    first = np.repeat(off[:n] + 49, nc) + 4 * (np.arange(m, dtype=np.int64) - np.repeat(reads.cig_off[a:b] - c0, nc))
                    buf[(first[:, None] + np.arange(4)[None, :]).ravel()] = words.ravel()
                w = np.nonzero(xs)[0]
                if len(w):
                    at = off[w] + 49 + 4 * nc[w]
                    buf[at] = ord("X"); buf[at + 1] = ord("S"); buf[at + 2] = ord("A")
                    buf[at + 3] = np.where(reads.rev[a:b][w] != 0, ord("-"), ord("+"))
                pending += buf.tobytes()
            last = b >= reads.n
            cut = len(pending) if last else (len(pending) // BLOCK) * BLOCK
            pieces = [pending[i:i + BLOCK] for i in range(0, cut, BLOCK)]
            pending = pending[cut:]
            for comp in pool.map(block, pieces):
                fh.write(comp)
            if last:
                break
        fh.write(block(b""))


def _compact_rows(vals: np.ndarray, mask: np.ndarray):
    """Row-wise compaction of a padded [R, L] array: returns flat values (row-major) and offsets."""
    cnt = mask.sum(axis=1)
    off = np.zeros(len(cnt) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    return vals[mask], off


def make_reads(anno: Annotation, n_reads: int, n_exons: int, seed: int, full_frac: float = 0.7,
               ont: bool = False, micro_exons: int = 0, xs_conflict_frac: float = 0.0,
               chunk: int = 1_000_000, unsorted: bool = False) -> Reads:
    """``n_reads`` alignments with about ``n_exons`` exons each against ``anno``.

    Full-length reads (``full_frac``) copy a whole transcript, the rest take ``n_exons`` consecutive
    exons; the mode mix of SURVEY.md section 8(d) is then applied.
    """
    rng = np.random.default_rng([seed, 0xBEAD])
    T = anno.n_tx
    tx_n = np.diff(anno.tx_ex_off).astype(np.int64)
    Lmax = int(tx_n.max())

    tx = rng.integers(0, T, size=n_reads)
    ntx = tx_n[tx]
    full = rng.random(n_reads) < full_frac
    m = np.where(full, ntx, np.minimum(ntx, n_exons))
    a = np.where(full, 0, (rng.random(n_reads) * (ntx - m + 1)).astype(np.int64))
    u = rng.random(n_reads)
    # 0 exact 55% | 1 ragged 20% | 2 skip 10% | 3 alt donor 7% | 4 shifted 4% | 5 single 4%
    mode = np.searchsorted(np.array([0.55, 0.75, 0.85, 0.92, 0.96]), u, side="right").astype(np.int8)
    mode[(mode == 2) & (m < 3)] = 0
    mode[(mode == 3) & (m < 2)] = 0
    single = mode == 5
    a = np.where(single, a + (rng.random(n_reads) * m).astype(np.int64), a)
    m = np.where(single, 1, m)
    k_sel = (rng.random(n_reads) * np.maximum(m - 2, 1)).astype(np.int64) + 1       # internal exon for skip
    k_don = (rng.random(n_reads) * np.maximum(m - 1, 1)).astype(np.int64)            # exon whose donor moves
    d_don = rng.integers(1, 10, size=n_reads)
    d_left = rng.integers(-40, 41, size=n_reads)
    d_right = rng.integers(-40, 41, size=n_reads)
    shift = np.where(mode == 4, 30_000 + rng.integers(0, 1000, size=n_reads), 0)

    first_start = anno.ex_start[anno.tx_ex_off[tx] + a].astype(np.int64)
    first_end = anno.ex_end[anno.tx_ex_off[tx] + a].astype(np.int64)
    ragged = (mode == 1) | single
    # keep the moved start inside the first exon
    d_left = np.where(ragged, np.minimum(d_left, first_end - first_start - 5), 0)
    pos1 = first_start + d_left + shift                                   # 1-based start of the read
    pos1 = np.maximum(pos1, 1)
    tid = anno.tx_tid[tx].astype(np.int32)

    if unsorted:
        # locally disordered: a few percent of the records trade places with a neighbour up to 30
        # records away, so the history-dependent cursors of the reference are exercised without
        # pushing the cursor past most of the input
        order = np.lexsort((pos1, tid))
        pick = np.nonzero(rng.random(n_reads) < 0.04)[0]
        other = np.minimum(pick + rng.integers(1, 31, size=len(pick)), n_reads - 1)
        for i, j in zip(pick.tolist(), other.tolist()):
            order[i], order[j] = order[j], order[i]
    else:
        order = np.lexsort((pos1, tid))
    tx, m, a, mode, k_sel, k_don, d_don, d_left, d_right, shift, pos1, tid, single = [
        v[order] for v in (tx, m, a, mode, k_sel, k_don, d_don, d_left, d_right, shift, pos1, tid, single)]

    rev = anno.tx_rev[tx].astype(np.uint8)
    has_xs = np.zeros(n_reads, np.uint8)
    flag_rev = rev.copy()
    if xs_conflict_frac > 0:
        conflict = rng.random(n_reads) < xs_conflict_frac
        has_xs[conflict] = 1
        flag_rev[conflict] ^= 1                       # XS:A carries the real strand, FLAG contradicts it

    cig_parts, off_parts = [], []
    total = 0
    cols = np.arange(Lmax)[None, :]
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        sl = slice(lo, hi)
        base = (anno.tx_ex_off[tx[sl]] + a[sl])[:, None]
        valid = cols < m[sl][:, None]
        gidx = np.where(valid, base + cols, 0)
        S = anno.ex_start[gidx].astype(np.int64)
        E = anno.ex_end[gidx].astype(np.int64)
        md = mode[sl]
        rows = np.arange(hi - lo)
        # ragged ends / single exon: move outer boundaries
        rg = (md == 1) | single[sl]
        lastc = m[sl] - 1
        S[rows, 0] += d_left[sl]
        dr = np.where(rg, np.maximum(d_right[sl], -(E[rows, lastc] - S[rows, lastc] - 5)), 0)
        E[rows, lastc] += dr
        # alt donor: exon k_don end += d (stay below next exon start)
        ad = md == 3
        kd = np.where(ad, k_don[sl], 0)
        room = S[rows, np.minimum(kd + 1, Lmax - 1)] - E[rows, kd] - 5
        E[rows, kd] += np.where(ad, np.minimum(d_don[sl], np.maximum(room, 0)), 0)
        # exon skip
        sk = md == 2
        valid[rows[sk], k_sel[sl][sk]] = False
        S += shift[sl][:, None]
        E += shift[sl][:, None]

        flatS, eoff = _compact_rows(S, valid)
        flatE, _ = _compact_rows(E, valid)
        nex = np.diff(eoff)
        if not ont and micro_exons == 0:
            # clean CIGAR: M N M N ... M   (c = 2n - 1)
            c = 2 * nex - 1
            coff = np.zeros(len(c) + 1, np.int64)
            np.cumsum(c, out=coff[1:])
            ops = np.empty(int(coff[-1]), np.uint32)
            first_of_read = np.repeat(coff[:-1], nex)
            kk = np.arange(len(flatS)) - np.repeat(eoff[:-1], nex)
            mpos = first_of_read + 2 * kk
            ops[mpos] = ((flatE - flatS + 1).astype(np.uint32) << 4) | CIG_M
            notlast = kk < np.repeat(nex, nex) - 1
            gap = np.empty(len(flatS), np.int64)
            gap[:-1] = flatS[1:] - flatE[:-1] - 1
            gap[-1] = 0
            ops[mpos[notlast] + 1] = (gap[notlast].astype(np.uint32) << 4) | CIG_N
        else:
            ops, coff = _noisy_cigars(rng, flatS, flatE, eoff, ont, micro_exons)
        cig_parts.append(ops)
        off_parts.append(coff[:-1] + total)
        total += int(coff[-1])

    cig = np.concatenate(cig_parts) if cig_parts else np.zeros(0, np.uint32)
    cig_off = np.concatenate(off_parts + [np.array([total], np.int64)])
    return Reads(anno.chrom_names, tid, (pos1 - 1).astype(np.int32), np.where(has_xs == 1, rev, flag_rev).astype(np.uint8),
                 flag_rev, has_xs, cig_off, cig, 0, 250_000_000, not unsorted)


def _noisy_cigars(rng, flatS, flatE, eoff, ont: bool, micro_exons: int):
    """ONT-like CIGARs: every exon is cut into M runs separated by short I/D (a few D > 50), up to
    ``micro_exons`` introns per read get a 1..6 bp micro-exon inserted, and reads get soft clips.
    Returns (ops uint32, offsets int64).  Loop is per read chunk but vectorised per exon."""
    n_reads = len(eoff) - 1
    nex = np.diff(eoff)
    exlen = (flatE - flatS + 1).astype(np.int64)
    # --- per exon: pieces
    if ont:
        step = rng.integers(10, 16, size=len(exlen))
        npiece = np.maximum(exlen // step, 1)                # M runs per exon
    else:
        npiece = np.ones(len(exlen), np.int64)
    nind = npiece - 1                                         # indels inside the exon
    # indel kinds/lengths, flat over all exons
    tot_ind = int(nind.sum())
    is_del = rng.random(tot_ind) < 0.5
    ind_len = np.where(is_del, rng.integers(1, 6, size=tot_ind), rng.integers(1, 4, size=tot_ind))
    big = rng.random(tot_ind) < 0.002
    ind_len = np.where(is_del & big, rng.integers(51, 90, size=tot_ind), ind_len)
    ind_ref = np.where(is_del, ind_len, 0)
    ind_exon = np.repeat(np.arange(len(exlen)), nind)
    del_per_exon = np.bincount(ind_exon, weights=ind_ref, minlength=len(exlen)).astype(np.int64)
    # an exon must keep >= 1 base per M run; drop deletions of exons that cannot afford them
    cant = del_per_exon + npiece > exlen
    bad = cant[ind_exon]
    ind_len = np.where(bad & is_del, 1, ind_len)
    is_del = is_del & ~bad
    ind_ref = np.where(is_del, ind_len, 0)
    del_per_exon = np.bincount(ind_exon, weights=ind_ref, minlength=len(exlen)).astype(np.int64)
    mbases = exlen - del_per_exon                             # bases to spread over npiece M runs
    piece_exon = np.repeat(np.arange(len(exlen)), npiece)
    piece_k = np.arange(len(piece_exon)) - np.repeat(np.cumsum(npiece) - npiece, npiece)
    basem = mbases[piece_exon] // npiece[piece_exon]
    extra = (piece_k < (mbases[piece_exon] % npiece[piece_exon])).astype(np.int64)
    mlen = basem + extra
    # ops per exon = npiece + nind, interleaved M I/D M ...
    ops_per_exon = npiece + nind
    # --- introns: gap after exon (not last of read); optional micro-exon
    kk = np.arange(len(flatS)) - np.repeat(eoff[:-1], nex)
    notlast = kk < np.repeat(nex, nex) - 1
    gap = np.zeros(len(flatS), np.int64)
    gap[:-1] = flatS[1:] - flatE[:-1] - 1
    gap[~notlast] = 0
    has_micro = np.zeros(len(flatS), bool)
    if micro_exons > 0:
        cand = notlast & (gap > 40)
        # choose up to micro_exons introns per read
        key = np.where(cand, rng.random(len(flatS)), 2.0)
        read_of = np.repeat(np.arange(n_reads), nex)
        order = np.lexsort((key, read_of))
        rank = np.empty(len(flatS), np.int64)
        rank[order] = np.arange(len(flatS)) - np.repeat(eoff[:-1], nex)
        has_micro = cand & (rank < micro_exons)
    mic_len = rng.integers(1, 7, size=len(flatS))
    g1 = np.where(has_micro, 10 + (rng.random(len(flatS)) * np.maximum(gap - mic_len - 20, 1)).astype(np.int64), 0)
    g2 = np.where(has_micro, gap - g1 - mic_len, 0)
    intron_ops = np.where(notlast, np.where(has_micro, 3, 1), 0)
    clip = 2 if ont else 0                                     # leading + trailing soft clip
    # --- lay out
    per_exon_total = ops_per_exon + intron_ops
    c = np.bincount(np.repeat(np.arange(n_reads), nex), weights=per_exon_total, minlength=n_reads).astype(np.int64) + clip
    coff = np.zeros(n_reads + 1, np.int64)
    np.cumsum(c, out=coff[1:])
    ops = np.zeros(int(coff[-1]), np.uint32)
    ex_first = np.cumsum(per_exon_total) - per_exon_total      # position of exon's first op, without clip shifts
    read_of = np.repeat(np.arange(n_reads), nex)
    ex_first = ex_first + (read_of * clip) + (1 if clip else 0)
    if clip:
        ops[coff[:-1]] = (rng.integers(1, 60, size=n_reads).astype(np.uint32) << 4) | CIG_S
        ops[coff[1:] - 1] = (rng.integers(1, 60, size=n_reads).astype(np.uint32) << 4) | CIG_S
    # M runs
    ops[ex_first[piece_exon] + 2 * piece_k] = (mlen.astype(np.uint32) << 4) | CIG_M
    # indels
    ind_k = np.arange(tot_ind) - np.repeat(np.cumsum(nind) - nind, nind)
    ops[ex_first[ind_exon] + 2 * ind_k + 1] = (ind_len.astype(np.uint32) << 4) | np.where(is_del, CIG_D, CIG_I).astype(np.uint32)
    # introns
    ipos = ex_first + ops_per_exon
    plain = notlast & ~has_micro
    ops[ipos[plain]] = (gap[plain].astype(np.uint32) << 4) | CIG_N
    mi = has_micro
    ops[ipos[mi]] = (g1[mi].astype(np.uint32) << 4) | CIG_N
    ops[ipos[mi] + 1] = (mic_len[mi].astype(np.uint32) << 4) | CIG_M
    ops[ipos[mi] + 2] = (g2[mi].astype(np.uint32) << 4) | CIG_N
    return ops, coff


# --------------------------------------------------------------------------- SJ table

@dataclass
class Junctions:
    chrom: List[str]          # per row chromosome name (may include names absent from the header)
    tid: np.ndarray           # int32 index of chrom in the BAM header order (rows sorted by tid, don, acc)
    don: np.ndarray           # int32 first intron base
    acc: np.ndarray           # int32 last intron base
    strand: np.ndarray        # 0/1/2 STAR code
    uniq: np.ndarray
    multi: np.ndarray

    def write(self, path: str) -> None:
        with open(path, "w") as fh:
            for i in range(len(self.don)):
                fh.write("%s\t%d\t%d\t%d\t1\t%d\t%d\t%d\t30\n" % (
                    self.chrom[i], self.don[i], self.acc[i], self.strand[i], 1 if self.uniq[i] > 3 else 0,
                    self.uniq[i], self.multi[i]))


def make_junctions(anno: Annotation, exon_off: np.ndarray, ex_start: np.ndarray, ex_end: np.ndarray,
                   read_tid: np.ndarray, seed: int, cover: float = 0.8, max_rows: int = 2_000_000) -> Junctions:
    """STAR-like table: every annotated junction plus ``cover`` of the junctions seen in the reads
    (``exon_off/ex_start/ex_end`` = exon chains of the reads, e.g. from the oracle)."""
    rng = np.random.default_rng([seed, 0x51])
    # annotation junctions
    a_n = np.diff(anno.tx_ex_off)
    kk = np.arange(anno.n_exons) - np.repeat(anno.tx_ex_off[:-1], a_n)
    nl = kk < np.repeat(a_n, a_n) - 1
    a_don = anno.ex_end[nl].astype(np.int64) + 1
    a_acc = anno.ex_start[np.nonzero(nl)[0] + 1].astype(np.int64) - 1
    a_tid = np.repeat(anno.tx_tid, a_n)[nl].astype(np.int64)
    # read junctions
    r_n = np.diff(exon_off)
    kk = np.arange(len(ex_start)) - np.repeat(exon_off[:-1], r_n)
    nl = kk < np.repeat(r_n, r_n) - 1
    r_don = ex_end[nl].astype(np.int64) + 1
    r_acc = ex_start[np.nonzero(nl)[0] + 1].astype(np.int64) - 1
    r_tid = np.repeat(read_tid, r_n)[nl].astype(np.int64)
    tid = np.concatenate([a_tid, r_tid]); don = np.concatenate([a_don, r_don]); acc = np.concatenate([a_acc, r_acc])
    key = np.stack([tid, don, acc], axis=1)
    key = np.unique(key, axis=0)
    keep = rng.random(len(key)) < cover
    key = key[keep][:max_rows]
    uniq = rng.integers(0, 25, size=len(key))
    multi = rng.integers(0, 4, size=len(key))
    chrom = [anno.chrom_names[int(t)] for t in key[:, 0]]
    return Junctions(chrom, key[:, 0].astype(np.int32), key[:, 1].astype(np.int32), key[:, 2].astype(np.int32),
                     rng.integers(1, 3, size=len(key)), uniq, multi)


def make_junctions_fast(anno: Annotation, exon_off: np.ndarray, ex_start: np.ndarray, ex_end: np.ndarray,
                        read_tid: np.ndarray, seed: int, cover: float = 0.8, max_rows: int = 0x7ffffff0) -> Junctions:
    """``make_junctions`` for tens of millions of read junctions (bench.py's second option set): the same table -- every
    annotated junction plus the junctions of the reads, ``cover`` of the distinct rows kept, sorted by (tid, don, acc) -- with
    the rows de-duplicated as one 64-bit key each (tid 8 bits | don 31 bits | intron length 25 bits) instead of a row-wise
    ``np.unique``.  Junctions whose intron does not fit 25 bits are left out (none in the generator's output)."""
    rng = np.random.default_rng([seed, 0x51])

    def keys(tid, n_per, starts, ends):
        n_per = np.asarray(n_per, np.int64)
        total = int(n_per.sum())
        first = np.zeros(len(n_per) + 1, np.int64)
        np.cumsum(n_per, out=first[1:])
        last = np.zeros(total, bool)
        last[first[1:][n_per > 0] - 1] = True
        nl = np.nonzero(~last)[0]                        # exons with a junction behind them
        don = ends[nl].astype(np.int64) + 1
        acc = starts[nl + 1].astype(np.int64) - 1
        t = np.repeat(np.asarray(tid, np.int64), n_per)[nl]
        ln = acc - don
        ok = (ln >= 0) & (ln < (1 << 25)) & (don >= 0) & (don < (1 << 31)) & (t >= 0) & (t < 256)
        return (t[ok] << 56) | (don[ok] << 25) | ln[ok]

    ka = keys(anno.tx_tid, np.diff(anno.tx_ex_off), anno.ex_start, anno.ex_end)
    kr = keys(read_tid, np.diff(exon_off), ex_start, ex_end)
    key = np.unique(np.concatenate([ka, kr]))
    keep = rng.random(len(key)) < cover
    key = key[keep][:max_rows]
    tid = (key >> 56).astype(np.int32)
    don = ((key >> 25) & ((1 << 31) - 1)).astype(np.int32)
    acc = (don.astype(np.int64) + (key & ((1 << 25) - 1))).astype(np.int32)
    uniq = rng.integers(0, 25, size=len(key))
    multi = rng.integers(0, 4, size=len(key))
    names = np.asarray(anno.chrom_names, dtype=object)
    chrom = list(names[tid]) if len(key) else []
    return Junctions(chrom, tid, don, acc, rng.integers(1, 3, size=len(key)), uniq, multi)
