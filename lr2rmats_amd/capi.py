"""ctypes view of the C-ABI in ``include/lr2rmats_hip.h`` (``libl2r_hip.so``).

This is plumbing for tests, bench.py and the multi-GPU driver; the product
host code is C (``lr2rmats_amd/host``) and links the same library.  There is no
fallback: if the shared library is missing or no GPU is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L2R_HIP_LIB") or os.path.join(HERE, "lib", "libl2r_hip.so")      # (L2R_HIP_LIB: diagnostics, tools/ab.py compares builds)

INFO_KNOWN, INFO_KNOWN_SITE, INFO_FULL, INFO_REV, INFO_UNREL, INFO_SJ_CHECKED, INFO_SJ_PASS, INFO_ACCEPTED = \
    1, 2, 4, 8, 16, 32, 64, 128
EXF_NOVEL_EXON, EXF_NOVEL_DON, EXF_NOVEL_ACC, EXF_NOVEL_JUNC, EXF_UNREL_JUNC = 1, 2, 4, 8, 16
WANT_RESULTS, WANT_ACCEPTED = 1, 2
N_STAGES = 8
STAGE_NAMES = ["pass_a", "scan_tiles", "classify_fast", "classify_generic", "validate_sj", "scan_accepted",
               "gather_accepted", "reserved"]

EXPORTS = [
    "l2r_abi_version", "l2r_last_error", "l2r_device_count", "l2r_create", "l2r_destroy", "l2r_set_params",
    "l2r_set_outputs", "l2r_set_annotation", "l2r_set_junctions", "l2r_upload_reads", "l2r_run", "l2r_sync", "l2r_run_timed",
    "l2r_result_sizes", "l2r_download", "l2r_download_accepted", "l2r_device_view_get", "l2r_stream", "l2r_classify",
    "l2r_stage_kernel", "l2r_set_annotation_cache", "l2r_annotation_cache_state", "l2r_filter_score", "l2r_filter_select",
    "l2r_debug_counters", "l2r_debug_stamps", "l2r_debug_tile_times", "l2r_upload_index_ms", "l2r_hint_single_run",
    "l2r_xchg_id_bytes", "l2r_xchg_unique_id", "l2r_xchg_create", "l2r_xchg_gather_results", "l2r_xchg_gather_accepted", "l2r_xchg_destroy",
]

_i32p, _i64p, _u8p, _u32p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)


class Params(C.Structure):
    _fields_ = [("min_exon", C.c_int32), ("min_intron", C.c_int32), ("max_delet", C.c_int32), ("ss_dis", C.c_int32),
                ("end_dis", C.c_int32), ("full_level", C.c_int32), ("split_trans", C.c_int32), ("use_multi", C.c_int32),
                ("min_sj_cnt", C.c_int32), ("force_strand", C.c_int32), ("single_exon_ovlp_frac", C.c_float)]


class CAnnotation(C.Structure):
    _fields_ = [("n_tx", C.c_int64), ("n_exon", C.c_int64), ("tx_tid", _i32p), ("tx_start", _i32p), ("tx_end", _i32p),
                ("tx_rev", _u8p), ("tx_ex_off", _i64p), ("ex_start", _i32p), ("ex_end", _i32p)]


class CJunctions(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", _i32p), ("don", _i32p), ("acc", _i32p), ("uniq_c", _i32p), ("multi_c", _i32p)]


class CReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_cigar", C.c_int64), ("tid", _i32p), ("pos", _i32p), ("rev", _u8p),
                ("cig_off", _i64p), ("cig", _u32p), ("first_read_index", C.c_int64), ("cig_summary", _u32p)]


class CResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("ex_cap", C.c_int64), ("n_exons", C.c_int64), ("ex_off", _i64p),
                ("ex_start", _i32p), ("ex_end", _i32p), ("ex_flag", _u8p), ("info", _u32p), ("ref_tx", _i32p)]


class CAccepted(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("ex_cap", C.c_int64), ("n_exons", C.c_int64), ("rec", C.c_void_p),
                ("ex_off", _i64p), ("ex_start", _i32p), ("ex_end", _i32p), ("ex_flag", _u8p)]


class CDeviceView(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_exons", C.c_int64), ("n_accepted", C.c_int64), ("n_accepted_exons", C.c_int64),
                ("ex_off", C.c_void_p), ("ex_start", C.c_void_p), ("ex_end", C.c_void_p), ("ex_flag", C.c_void_p),
                ("info", C.c_void_p), ("ref_tx", C.c_void_p), ("acc_rec", C.c_void_p), ("acc_ex_off", C.c_void_p),
                ("acc_ex_start", C.c_void_p), ("acc_ex_end", C.c_void_p), ("acc_ex_flag", C.c_void_p)]


class CFilterParams(C.Structure):
    _fields_ = [("cov_rate", C.c_float), ("map_qual", C.c_float), ("sec_rat", C.c_float), ("min_intron_n", C.c_int32)]


class CFilterRecords(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_cigar", C.c_int64), ("flag", C.c_void_p), ("tid", C.c_void_p), ("pos", C.c_void_p),
                ("l_qseq", C.c_void_p), ("nm", C.c_void_p), ("cig_off", C.c_void_p), ("cig", C.c_void_p)]


class CFilterSpans(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p)]


class CTiming(C.Structure):
    _fields_ = [("stage_ms", C.c_float * N_STAGES), ("total_ms", C.c_float), ("iters", C.c_int32)]


def accepted_mask(info: np.ndarray, has_sj: bool, split: bool) -> np.ndarray:
    """Which reads check_trans() hands to novel_T / merge_trans (src/update_gtf.c:943-960), from the other info bits:
    full, not known, has a known site; with a junction table additionally the junction check passed, or -s is set
    (the read then goes on as split pieces).  The engine stores this as INFO_ACCEPTED."""
    cand = (info & (INFO_FULL | INFO_KNOWN | INFO_KNOWN_SITE)) == (INFO_FULL | INFO_KNOWN_SITE)
    if not has_sj:
        return cand
    return cand & (((info & INFO_SJ_PASS) != 0) | bool(split))


ACC_REC_DTYPE = np.dtype([("read_lo", "<u4"), ("read_hi", "<u4"), ("info", "<u4"), ("ref_tx", "<i4")])


def default_params(**kw) -> Params:
    """Defaults of reference src/update_gtf.c:24-35."""
    p = Params(3, 3, 50, 0, 0x7fffffff, 5, 0, 0, 1, 0, 0.80)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


_lib = None


def load_library():
    """Load libl2r_hip.so (raises OSError if it was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: build it with `make -C lr2rmats_amd/csrc` (no CPU fallback exists)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.l2r_last_error.restype = C.c_char_p
        lib.l2r_create.restype = C.c_void_p
        lib.l2r_create.argtypes = [C.c_int]
        lib.l2r_destroy.argtypes = [C.c_void_p]
        lib.l2r_stream.restype = C.c_void_p
        lib.l2r_stream.argtypes = [C.c_void_p]
        for name in ("l2r_set_params", "l2r_set_outputs", "l2r_set_annotation", "l2r_set_junctions", "l2r_upload_reads", "l2r_download",
                     "l2r_download_accepted", "l2r_device_view_get"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
        lib.l2r_set_outputs.argtypes = [C.c_void_p, C.c_uint]
        lib.l2r_run.argtypes = [C.c_void_p]
        lib.l2r_sync.argtypes = [C.c_void_p]
        lib.l2r_run_timed.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.l2r_result_sizes.argtypes = [C.c_void_p, _i64p, _i64p, _i64p, _i64p]
        lib.l2r_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.l2r_set_annotation_cache.argtypes = [C.c_void_p, C.c_char_p]
        lib.l2r_annotation_cache_state.argtypes = [C.c_void_p]
        lib.l2r_filter_score.argtypes = [C.c_void_p] * 7
        lib.l2r_filter_select.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5
        lib.l2r_hint_single_run.argtypes = [C.c_void_p, C.c_int]
        lib.l2r_upload_index_ms.restype = C.c_float
        lib.l2r_upload_index_ms.argtypes = [C.c_void_p]
        lib.l2r_stage_kernel.restype = C.c_char_p
        lib.l2r_stage_kernel.argtypes = [C.c_void_p, C.c_int]
        _lib = lib
    return _lib


class L2RError(RuntimeError):
    pass


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


@dataclass
class Result:
    ex_off: np.ndarray
    ex_start: np.ndarray
    ex_end: np.ndarray
    ex_flag: np.ndarray
    info: np.ndarray
    ref_tx: np.ndarray


@dataclass
class Accepted:
    rec: np.ndarray          # structured ACC_REC_DTYPE
    ex_off: np.ndarray
    ex_start: np.ndarray
    ex_end: np.ndarray
    ex_flag: np.ndarray

    @property
    def read_index(self) -> np.ndarray:
        return self.rec["read_lo"].astype(np.int64) | (self.rec["read_hi"].astype(np.int64) << 32)


class Engine:
    """One GPU context (one process per GPU)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.l2r_create(device)
        if not self.ctx:
            raise L2RError(self.lib.l2r_last_error().decode())
        self._keep = {}

    def close(self):
        if self.ctx:
            self.lib.l2r_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise L2RError("rc=%d: %s" % (rc, self.lib.l2r_last_error().decode()))

    def set_params(self, p: Params):
        self._chk(self.lib.l2r_set_params(self.ctx, C.byref(p)))

    def set_outputs(self, want: int):
        """WANT_RESULTS (1) and/or WANT_ACCEPTED (2): include/lr2rmats_hip.h l2r_set_outputs."""
        self._chk(self.lib.l2r_set_outputs(self.ctx, want))

    def set_annotation(self, tx_tid, tx_start, tx_end, tx_rev, tx_ex_off, ex_start, ex_end):
        a = [np.ascontiguousarray(tx_tid, np.int32), np.ascontiguousarray(tx_start, np.int32),
             np.ascontiguousarray(tx_end, np.int32), np.ascontiguousarray(tx_rev, np.uint8),
             np.ascontiguousarray(tx_ex_off, np.int64), np.ascontiguousarray(ex_start, np.int32),
             np.ascontiguousarray(ex_end, np.int32)]
        ca = CAnnotation(len(a[0]), len(a[5]), _ptr(a[0], _i32p), _ptr(a[1], _i32p), _ptr(a[2], _i32p), _ptr(a[3], _u8p),
                         _ptr(a[4], _i64p), _ptr(a[5], _i32p), _ptr(a[6], _i32p))
        self._chk(self.lib.l2r_set_annotation(self.ctx, C.byref(ca)))

    def set_junctions(self, sj):
        if sj is None or len(sj[0]) == 0:
            self._chk(self.lib.l2r_set_junctions(self.ctx, None))
            return
        a = [np.ascontiguousarray(x, np.int32) for x in sj]
        cj = CJunctions(len(a[0]), *[_ptr(x, _i32p) for x in a])
        self._chk(self.lib.l2r_set_junctions(self.ctx, C.byref(cj)))

    # ---- `filter` (include/lr2rmats_hip.h: l2r_filter_score / l2r_filter_select)
    def filter_score(self, flag, tid, pos, l_qseq, nm, cig_off, cig, prm: "CFilterParams", spans=None):
        """(drop, score, intron_n) of every record; spans = (tid, start, end) of the -r transcripts in file order."""
        a = [np.ascontiguousarray(flag, np.uint16), np.ascontiguousarray(tid, np.int32), np.ascontiguousarray(pos, np.int32),
             np.ascontiguousarray(l_qseq, np.int32), np.ascontiguousarray(nm, np.int32), np.ascontiguousarray(cig_off, np.int64),
             np.ascontiguousarray(cig, np.uint32)]
        n = int(a[0].shape[0])
        recs = CFilterRecords(n, int(a[6].shape[0]), *[x.ctypes.data for x in a])
        sp = None
        if spans is not None:
            s = [np.ascontiguousarray(x, np.int32) for x in spans]
            sp = CFilterSpans(int(s[0].shape[0]), *[x.ctypes.data for x in s])
        drop = np.zeros(max(n, 1), np.uint8); score = np.zeros(max(n, 1), np.int32); intron = np.zeros(max(n, 1), np.int32)
        self._chk(self.lib.l2r_filter_score(self.ctx, C.byref(recs), C.byref(prm), C.byref(sp) if sp is not None else None,
                                            drop.ctypes.data, score.ctypes.data, intron.ctypes.data))
        return drop[:n], score[:n], intron[:n]

    def filter_select(self, group_off, score, intron_n, prm: "CFilterParams"):
        g = np.ascontiguousarray(group_off, np.int64); s = np.ascontiguousarray(score, np.int32); i = np.ascontiguousarray(intron_n, np.int32)
        ng = int(g.shape[0]) - 1
        win = np.zeros(max(ng, 1), np.int64)
        self._chk(self.lib.l2r_filter_select(self.ctx, ng, g.ctypes.data, s.ctypes.data, i.ctypes.data, C.byref(prm), win.ctypes.data))
        return win[:ng]

    def set_annotation_cache(self, directory) -> None:
        """Keep the annotation tables on disk under ``directory`` (None: off); see include/lr2rmats_hip.h."""
        self._chk(self.lib.l2r_set_annotation_cache(self.ctx, directory.encode() if directory else None))

    def annotation_cache_state(self) -> int:
        """0 no cache, 1 built and stored, 2 read from the cache (last set_annotation)."""
        return int(self.lib.l2r_annotation_cache_state(self.ctx))

    def upload_reads(self, tid, pos, rev, cig_off, cig, first_read_index: int = 0, cig_summary=None):
        """``cig_summary``: the reader's per-record CIGAR summaries ([N, 3] uint32, ``synth.cigar_summary``; None: the engine walks the CIGARs)."""
        a = [np.ascontiguousarray(tid, np.int32), np.ascontiguousarray(pos, np.int32), np.ascontiguousarray(rev, np.uint8),
             np.ascontiguousarray(cig_off, np.int64), np.ascontiguousarray(cig, np.uint32)]
        sm = None if cig_summary is None else np.ascontiguousarray(cig_summary, np.uint32)
        if sm is not None and sm.shape != (len(a[0]), 3):
            raise ValueError("cig_summary: expected shape (n_reads, 3)")
        cr = CReads(len(a[0]), len(a[4]), _ptr(a[0], _i32p), _ptr(a[1], _i32p), _ptr(a[2], _u8p), _ptr(a[3], _i64p),
                    _ptr(a[4], _u32p), first_read_index, _ptr(sm, _u32p) if sm is not None else None)
        self._chk(self.lib.l2r_upload_reads(self.ctx, C.byref(cr)))

    def hint_single_run(self, on: bool = True):
        """Every following upload will be classified once (l2r_hint_single_run): no tile index, the two-kernel pipeline."""
        self._chk(self.lib.l2r_hint_single_run(self.ctx, 1 if on else 0))

    def upload_index_ms(self) -> float:
        """GPU time of the last upload's tile index (k_tile_index), ms."""
        return float(self.lib.l2r_upload_index_ms(self.ctx))

    def run(self):
        self._chk(self.lib.l2r_run(self.ctx))

    def sync(self):
        self._chk(self.lib.l2r_sync(self.ctx))

    def run_timed(self, iters: int) -> dict:
        t = CTiming()
        self._chk(self.lib.l2r_run_timed(self.ctx, iters, C.byref(t)))
        return {"total_ms": float(t.total_ms), "iters": int(t.iters),
                "stage_ms": {STAGE_NAMES[i]: float(t.stage_ms[i]) for i in range(N_STAGES - 1)},
                # the kernel behind every stage for the pipeline the engine chose for these records (slab / classic)
                "kernel_ms": {(self.lib.l2r_stage_kernel(self.ctx, i) or b"").decode(): float(t.stage_ms[i])
                              for i in range(N_STAGES - 1) if self.lib.l2r_stage_kernel(self.ctx, i)}}

    def sizes(self):
        v = [C.c_int64(0) for _ in range(4)]
        self._chk(self.lib.l2r_result_sizes(self.ctx, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def download(self) -> Result:
        n, x, _, _ = self.sizes()
        off = np.zeros(n + 1, np.int64); s = np.zeros(max(x, 1), np.int32); e = np.zeros(max(x, 1), np.int32)
        f = np.zeros(max(x, 1), np.uint8); info = np.zeros(max(n, 1), np.uint32); ref = np.zeros(max(n, 1), np.int32)
        cr = CResult(n, max(x, 1), 0, _ptr(off, _i64p), _ptr(s, _i32p), _ptr(e, _i32p), _ptr(f, _u8p), _ptr(info, _u32p),
                     _ptr(ref, _i32p))
        self._chk(self.lib.l2r_download(self.ctx, C.byref(cr)))
        return Result(off, s[:x], e[:x], f[:x], info[:n], ref[:n])

    def download_accepted(self) -> Accepted:
        _, _, m, x = self.sizes()
        rec = np.zeros(max(m, 1), ACC_REC_DTYPE); off = np.zeros(m + 1, np.int64)
        s = np.zeros(max(x, 1), np.int32); e = np.zeros(max(x, 1), np.int32); f = np.zeros(max(x, 1), np.uint8)
        ca = CAccepted(max(m, 1), max(x, 1), 0, rec.ctypes.data, _ptr(off, _i64p), _ptr(s, _i32p), _ptr(e, _i32p), _ptr(f, _u8p))
        self._chk(self.lib.l2r_download_accepted(self.ctx, C.byref(ca)))
        return Accepted(rec[:m], off, s[:x], e[:x], f[:x])

    def device_view(self) -> CDeviceView:
        v = CDeviceView()
        self._chk(self.lib.l2r_device_view_get(self.ctx, C.byref(v)))
        return v

    def classify(self, reads, params: Optional[Params] = None, first_read_index: int = 0) -> Result:
        """upload + run + sync + download for a ``synth.Reads``-like object."""
        if params is not None:
            self.set_params(params)
        self.upload_reads(reads.tid, reads.pos, reads.rev, reads.cig_off, reads.cig, first_read_index, getattr(reads, "cig_summary", None))
        self.run()
        self.sync()
        return self.download()
