"""CPU: the C-ABI library loads and exports every symbol include/lr2rmats_hip.h declares (no compute)."""
import ctypes
import os
import re

from lr2rmats_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lr2rmats_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(l2r_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(capi.EXPORTS)


def test_library_exports_every_symbol():
    assert os.path.exists(capi.LIB_PATH), "build with make -C lr2rmats_amd/csrc"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.l2r_abi_version.restype = ctypes.c_int
    assert lib.l2r_abi_version() == 3


def test_struct_layouts():
    assert ctypes.sizeof(capi.Params) == 44
    assert ctypes.sizeof(capi.CReads) == 72          # l2r_reads: + cig_summary (ABI 3)
    assert capi.ACC_REC_DTYPE.itemsize == 16
    assert ctypes.sizeof(capi.CTiming) == 4 * capi.N_STAGES + 8


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    lib = capi.load_library()
    assert not lib.l2r_create(0)
    assert b"no CPU path" in lib.l2r_last_error() or b"HIP" in lib.l2r_last_error()
