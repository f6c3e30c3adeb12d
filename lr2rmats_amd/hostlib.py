"""ctypes view of ``libl2r_host.so`` -- the C host side in its staged form.

``Job.open(argv)`` parses the update-gtf options and reads all inputs (C code),
``job.views()`` exposes the structure-of-arrays the engine consumes as numpy
arrays (zero copy), ``job.finish(result)`` runs the sequential tail and writes
every output file.  Used by the one-process-per-GPU driver (``dist.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

import numpy as np

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L2R_HOST_LIB") or os.path.join(HERE, "lib", "libl2r_host.so")      # (L2R_HOST_LIB: the sanitizer tests load libl2r_host_asan.so)
CLI_PATH = os.path.join(HERE, "bin", "lr2rmats")

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: build it with `make -C lr2rmats_amd/host`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.h_job_open.restype = C.c_void_p
        lib.h_job_open.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        lib.h_job_open2.restype = C.c_void_p
        lib.h_job_open2.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int]
        lib.h_job_open_rank.restype = C.c_void_p
        lib.h_job_open_rank.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        lib.h_job_shard.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.h_job_shard.restype = C.c_int
        lib.h_job_shard_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.h_job_shard_blocks.restype = None
        lib.h_job_views.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.h_job_finish.argtypes = [C.c_void_p, C.c_void_p]
        lib.h_job_finish.restype = C.c_int
        lib.h_job_free.argtypes = [C.c_void_p]
        lib.h_job_part_last_gene.argtypes = [C.c_void_p, C.c_int]
        lib.h_job_part_last_gene.restype = C.c_char_p
        lib.h_job_part_has_first_gene.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        lib.h_job_part_has_first_gene.restype = C.c_int
        lib.h_records_to_bam.argtypes = [C.c_char_p, C.c_char_p]
        lib.h_records_to_bam.restype = C.c_int
        lib.h_job_needs_all_reads.argtypes = [C.c_void_p]
        lib.h_job_needs_all_reads.restype = C.c_int
        lib.h_job_finish_accepted.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.h_job_finish_accepted.restype = C.c_int
        lib.h_job_finish_part.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]
        lib.h_job_finish_part.restype = C.c_int
        lib.h_job_out_path.argtypes = [C.c_void_p, C.c_int]
        lib.h_job_out_path.restype = C.c_char_p
        lib.h_job_write_summary.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_char_p]
        lib.h_job_write_summary.restype = C.c_int
        lib.h_job_open_outputs.argtypes = [C.c_void_p]
        lib.h_job_set_out_path.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        lib.h_cigar_summaries.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.h_cigar_summaries.restype = None
        _lib = lib
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype) if False else np.frombuffer(
        (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents)), dtype=dtype, count=n)


def cigar_summaries(cig_off, cig) -> np.ndarray:
    """The reader's per-record CIGAR summaries (l2r_reads::cig_summary, [N, 3] uint32) of records that are in memory already: the C loop of
    host/aln_reader.c (``synth.cigar_summary`` is the numpy form of the same rule, which the tests hold this to)."""
    off = np.ascontiguousarray(cig_off, np.int64)
    cg = np.ascontiguousarray(cig, np.uint32)
    out = np.empty((len(off) - 1, 3), np.uint32)
    load_library().h_cigar_summaries(len(off) - 1, off.ctypes.data, cg.ctypes.data, out.ctypes.data)
    return out

class Job:
    """One ``update-gtf`` invocation: argv = ["update-gtf", options..., in.bam, old.gtf]."""

    def __init__(self, argv: List[str], open_outputs: bool = True, rank: int = 0, world: int = 1):
        """``rank`` / ``world`` > 1: a rank of a one-process-per-GPU run -- for a coordinate-sorted BAM on the partitioned
        route only the rank's chromosome-aligned shard of the records is loaded (``shard()``)."""
        self.lib = load_library()
        args = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        rc = C.c_int(0)
        self._argv_keep = args
        if world > 1:
            self.h = self.lib.h_job_open_rank(len(argv), args, C.byref(rc), 1 if open_outputs else 0, rank, world)
        else:
            self.h = self.lib.h_job_open2(len(argv), args, C.byref(rc), 1 if open_outputs else 0)
        self.exit_code = rc.value
        if not self.h:
            raise SystemExit(self.exit_code or 1)
        self.prm = capi.Params()
        self.anno = capi.CAnnotation()
        self.sj = capi.CJunctions()
        self.reads = capi.CReads()
        self.lib.h_job_views(self.h, C.byref(self.prm), C.byref(self.anno), C.byref(self.sj), C.byref(self.reads))

    def shard(self):
        """(sharded, lo, hi, n_total): with ``sharded`` the read arrays hold the records [lo, hi) of the file only."""
        lo, hi, n = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        s = self.lib.h_job_shard(self.h, C.byref(lo), C.byref(hi), C.byref(n))
        return int(s), int(lo.value), int(hi.value), int(n.value)

    def shard_blocks(self):
        """shard()[0] == 2: [start block offset, offset in it, end block offset, offset in it, compressed bytes inflated, file size,
        records] of the rank's block range (host/aln_reader.c h_read_alignments_blocks)."""
        info = (C.c_int64 * 8)()
        self.lib.h_job_shard_blocks(self.h, info)
        return [int(x) for x in info[:7]]

    # numpy views (zero copy, valid until close())
    def annotation_arrays(self):
        a = self.anno
        t, e = a.n_tx, a.n_exon
        return dict(tx_tid=_arr(a.tx_tid, t, np.int32), tx_start=_arr(a.tx_start, t, np.int32), tx_end=_arr(a.tx_end, t, np.int32),
                    tx_rev=_arr(a.tx_rev, t, np.uint8), tx_ex_off=_arr(a.tx_ex_off, t + 1, np.int64),
                    ex_start=_arr(a.ex_start, e, np.int32), ex_end=_arr(a.ex_end, e, np.int32))

    def junction_arrays(self):
        s = self.sj
        if s.n == 0:
            return None
        return tuple(_arr(p, s.n, np.int32) for p in (s.tid, s.don, s.acc, s.uniq_c, s.multi_c))

    def read_arrays(self):
        r = self.reads
        # (cig_summary: the reader's per-record CIGAR summaries, l2r_reads::cig_summary; None for `-m g` input)
        sm = _arr(r.cig_summary, 3 * r.n_reads, np.uint32).reshape(-1, 3) if r.cig_summary else None
        return dict(tid=_arr(r.tid, r.n_reads, np.int32), pos=_arr(r.pos, r.n_reads, np.int32), rev=_arr(r.rev, r.n_reads, np.uint8),
                    cig_off=_arr(r.cig_off, r.n_reads + 1, np.int64), cig=_arr(r.cig, r.n_cigar, np.uint32), cig_summary=sm)

    def finish(self, ex_off, ex_start, ex_end, ex_flag, info, ref_tx) -> int:
        """Sequential tail + writers.  Arrays as in ``capi.Result`` (info carries the exon count in bits 8..31)."""
        n = int(info.shape[0])
        x = int(ex_start.shape[0])
        keep = [np.ascontiguousarray(ex_off, np.int64), np.ascontiguousarray(ex_start, np.int32), np.ascontiguousarray(ex_end, np.int32),
                np.ascontiguousarray(ex_flag, np.uint8), np.ascontiguousarray(info, np.uint32), np.ascontiguousarray(ref_tx, np.int32)]
        res = capi.CResult(n, x, x, keep[0].ctypes.data_as(capi._i64p), keep[1].ctypes.data_as(capi._i32p), keep[2].ctypes.data_as(capi._i32p),
                           keep[3].ctypes.data_as(capi._u8p), keep[4].ctypes.data_as(capi._u32p), keep[5].ctypes.data_as(capi._i32p))
        return self.lib.h_job_finish(self.h, C.byref(res))

    def needs_all_reads(self) -> bool:
        """An output of this run lists every read (detail.txt, -a / -k / -u, summary.txt)."""
        return bool(self.lib.h_job_needs_all_reads(self.h))

    def finish_accepted(self, read_idx, ex_off, ex_start, ex_end, ex_flag, info, ref_tx) -> int:
        """Tail + writers with the rows of the accepted reads only (input order; read_idx[k] = input index of row k)."""
        n, x = int(info.shape[0]), int(ex_start.shape[0])
        keep = [np.ascontiguousarray(ex_off, np.int64), np.ascontiguousarray(ex_start, np.int32), np.ascontiguousarray(ex_end, np.int32),
                np.ascontiguousarray(ex_flag, np.uint8), np.ascontiguousarray(info, np.uint32), np.ascontiguousarray(ref_tx, np.int32)]
        idx = np.ascontiguousarray(read_idx, np.int64)
        res = capi.CResult(n, x, x, keep[0].ctypes.data_as(capi._i64p), keep[1].ctypes.data_as(capi._i32p), keep[2].ctypes.data_as(capi._i32p),
                           keep[3].ctypes.data_as(capi._u8p), keep[4].ctypes.data_as(capi._u32p), keep[5].ctypes.data_as(capi._i32p))
        return self.lib.h_job_finish_accepted(self.h, C.byref(res), idx.ctypes.data_as(C.POINTER(C.c_int64)))

    # ---- partitioned tail (shards cut at chromosome boundaries; see l2r_host.h h_job_finish_part)
    N_SUMMARY = 16
    OUTPUTS = ("gtf", "bed", "bam_gtf", "detail", "known", "novel", "unrecog", "summary")

    def open_outputs(self) -> None:
        self.lib.h_job_open_outputs(self.h)

    def set_out_path(self, which: int, path: str) -> None:
        self.lib.h_job_set_out_path(self.h, which, path.encode())

    def out_path(self, which: int):
        p = self.lib.h_job_out_path(self.h, which)
        return p.decode() if p else None

    def finish_part(self, lo: int, hi: int, ex_off, ex_start, ex_end, ex_flag, info, ref_tx, suffix: str, stdout_base: str,
                    first_part: bool) -> np.ndarray:
        """Tail + writers for the reads [lo, hi) into "<output><suffix>" files; returns the summary counters."""
        n, x = int(info.shape[0]), int(ex_start.shape[0])
        keep = [np.ascontiguousarray(ex_off, np.int64), np.ascontiguousarray(ex_start, np.int32), np.ascontiguousarray(ex_end, np.int32),
                np.ascontiguousarray(ex_flag, np.uint8), np.ascontiguousarray(info, np.uint32), np.ascontiguousarray(ref_tx, np.int32)]
        res = capi.CResult(n, x, x, keep[0].ctypes.data_as(capi._i64p), keep[1].ctypes.data_as(capi._i32p), keep[2].ctypes.data_as(capi._i32p),
                           keep[3].ctypes.data_as(capi._u8p), keep[4].ctypes.data_as(capi._u32p), keep[5].ctypes.data_as(capi._i32p))
        cnt = np.zeros(self.N_SUMMARY, np.int64)
        self.lib.h_job_finish_part(self.h, lo, hi, C.byref(res), suffix.encode(), stdout_base.encode(), 1 if first_part else 0,
                                   cnt.ctypes.data_as(C.POINTER(C.c_int64)))
        return cnt

    # The two gene lists of summary.txt (genes of the updated transcripts, genes of the known reads) are the one place where
    # the reference looks across a chromosome boundary: the last entry's gene_id is compared before the tid break
    # (l2r_host.h h_part_genes).  After finish_part(): the part's last ids, and whether an id was added under its first tid.
    GENE_COUNTERS = (0, 7)

    def part_last_genes(self):
        out = []
        for q in range(2):
            g = self.lib.h_job_part_last_gene(self.h, q)
            out.append(g.decode() if g is not None else None)
        return out

    def part_has_first_gene(self, which: int, gid) -> bool:
        return gid is not None and bool(self.lib.h_job_part_has_first_gene(self.h, which, gid.encode()))

    def write_summary(self, counters: np.ndarray, path: str) -> None:
        c = np.ascontiguousarray(counters, np.int64)
        self.lib.h_job_write_summary(self.h, c.ctypes.data_as(C.POINTER(C.c_int64)), path.encode())

    def close(self):
        if self.h:
            self.lib.h_job_free(self.h)
            self.h = None


def records_to_bam(in_path: str, out_path: str) -> int:
    """Every alignment record of ``in_path`` (SAM / gzip SAM / BAM) written as a BAM file: the reader, SAM->BAM encoder and
    BGZF writer of ``lr2rmats filter`` without its tests (no GPU)."""
    return load_library().h_records_to_bam(in_path.encode(), out_path.encode())


def run_cli(args, stdout_path=None, cwd=None, env=None) -> subprocess.CompletedProcess:
    """Run the C binary ``lr2rmats <args>`` (needs a GPU for update-gtf / bam2gtf / unique-gtf -m b).
    ``env``: extra environment variables (L2R_CHUNK_READS, L2R_ROUTE, L2R_THREADS ...)."""
    full_env = None
    if env:
        full_env = dict(os.environ)
        full_env.update({k: str(v) for k, v in env.items()})
    if stdout_path:
        with open(stdout_path, "wb") as fh:
            return subprocess.run([CLI_PATH] + list(args), stdout=fh, stderr=subprocess.PIPE, cwd=cwd, env=full_env)
    return subprocess.run([CLI_PATH] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, env=full_env)
