"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Import this only from tests/, from ``__graft_entry__.smoke()`` and from the
``cpu_baseline`` leg of bench.py.  Nothing under ``lr2rmats_amd/`` may import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "liboracle.so")
CLI = os.path.join(BUILD, "lr2rmats_oracle")

INFO_KNOWN, INFO_KNOWN_SITE, INFO_FULL, INFO_REV, INFO_UNREL, INFO_SJ_CHECKED, INFO_SJ_PASS = 1, 2, 4, 8, 16, 32, 64
EXF_NOVEL_EXON, EXF_NOVEL_DON, EXF_NOVEL_ACC, EXF_NOVEL_JUNC, EXF_UNREL_JUNC = 1, 2, 4, 8, 16


def build(force: bool = False) -> None:
    if force or not (os.path.exists(LIB) and os.path.exists(CLI)):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)


class Params(C.Structure):
    _fields_ = [("min_exon", C.c_int32), ("min_intron", C.c_int32), ("max_delet", C.c_int32),
                ("ss_dis", C.c_int32), ("end_dis", C.c_int32), ("full_level", C.c_int32),
                ("split_trans", C.c_int32), ("use_multi", C.c_int32), ("min_sj_cnt", C.c_int32),
                ("force_strand", C.c_int32), ("single_exon_ovlp_frac", C.c_float)]


def default_params(**kw) -> Params:
    p = Params(3, 3, 50, 0, 0x7fffffff, 5, 0, 0, 1, 0, 0.80)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@dataclass
class Result:
    ex_off: np.ndarray     # int64 [N+1]
    ex_start: np.ndarray   # int32
    ex_end: np.ndarray     # int32
    ex_flag: np.ndarray    # uint8
    info: np.ndarray       # uint32 [N]
    ref_tx: np.ndarray     # int32 [N]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.orc_classify_soa.restype = C.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None and a.size else C.cast(None, C.POINTER(t))


def classify_soa(r_tid, r_pos, r_rev, cig_off, cig,
                 tx_tid, tx_start, tx_end, tx_rev, tx_ex_off, ex_start, ex_end,
                 sj=None, params: Params | None = None) -> Result:
    """Run the sequential restatement on structure-of-arrays inputs (annotation in file order)."""
    params = params or default_params()
    n = int(r_tid.shape[0])
    cap = int(cig.shape[0]) + n + 1
    out_off = np.zeros(n + 1, np.int64)
    out_s = np.zeros(cap, np.int32)
    out_e = np.zeros(cap, np.int32)
    out_f = np.zeros(cap, np.uint8)
    out_info = np.zeros(n, np.uint32)
    out_ref = np.zeros(n, np.int32)
    c32 = lambda a: np.ascontiguousarray(a, np.int32)
    c64 = lambda a: np.ascontiguousarray(a, np.int64)
    c8 = lambda a: np.ascontiguousarray(a, np.uint8)
    r_tid, r_pos, r_rev = c32(r_tid), c32(r_pos), c8(r_rev)
    cig_off, cig = c64(cig_off), np.ascontiguousarray(cig, np.uint32)
    tx_tid, tx_start, tx_end, tx_rev = c32(tx_tid), c32(tx_start), c32(tx_end), c8(tx_rev)
    tx_ex_off, ex_start, ex_end = c64(tx_ex_off), c32(ex_start), c32(ex_end)
    if sj is None:
        n_sj = 0
        s_tid = s_don = s_acc = s_u = s_m = np.zeros(0, np.int32)
    else:
        s_tid, s_don, s_acc, s_u, s_m = [c32(x) for x in sj]
        n_sj = int(s_tid.shape[0])
    used = lib().orc_classify_soa(
        C.c_int64(n), _p(r_tid, C.c_int32), _p(r_pos, C.c_int32), _p(r_rev, C.c_uint8),
        _p(cig_off, C.c_int64), _p(cig, C.c_uint32),
        C.c_int64(int(tx_tid.shape[0])), _p(tx_tid, C.c_int32), _p(tx_start, C.c_int32), _p(tx_end, C.c_int32),
        _p(tx_rev, C.c_uint8), _p(tx_ex_off, C.c_int64), _p(ex_start, C.c_int32), _p(ex_end, C.c_int32),
        C.c_int64(n_sj), _p(s_tid, C.c_int32), _p(s_don, C.c_int32), _p(s_acc, C.c_int32),
        _p(s_u, C.c_int32), _p(s_m, C.c_int32),
        C.byref(params),
        C.c_int64(cap), _p(out_off, C.c_int64), _p(out_s, C.c_int32), _p(out_e, C.c_int32),
        _p(out_f, C.c_uint8), _p(out_info, C.c_uint32), _p(out_ref, C.c_int32))
    if used < 0:
        raise RuntimeError("oracle failed: %d" % used)
    return Result(out_off, out_s[:used].copy(), out_e[:used].copy(), out_f[:used].copy(), out_info, out_ref)


def run_cli(args, stdout_path=None, cwd=None) -> int:
    """Run ``lr2rmats_oracle <args>``; stdout to ``stdout_path`` if given. Returns the exit code."""
    build()
    if stdout_path:
        with open(stdout_path, "wb") as fh:
            return subprocess.run([CLI] + list(args), stdout=fh, stderr=subprocess.PIPE, cwd=cwd).returncode
    return subprocess.run([CLI] + list(args), stderr=subprocess.PIPE, cwd=cwd).returncode
