"""Shared helpers of the test-suite (host side)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402
from dwarf_p_cloudsc2_tl_ad_amd import binding as B  # noqa: E402
from oracle import refcall  # noqa: E402

HOSTCHECK_DIR = os.path.join(ROOT, "tests", "hostcheck")
HOSTCHECK_LIB = os.path.join(HOSTCHECK_DIR, "libhostcheck_sp.so" if B.SINGLE else "libhostcheck.so")


def build_hostcheck() -> str:
    if os.environ.get("CLOUDSC2_HOSTCHECK_LIB"):  # a differently built copy, e.g. the AddressSanitizer build of test_sanitize.py
        return os.environ["CLOUDSC2_HOSTCHECK_LIB"]
    src = os.path.join(HOSTCHECK_DIR, "hostcheck.hip")
    deps = [src] + [os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "csrc", f) for f in ("cloudsc2_level.hpp", "cloudsc2_column.hpp")]
    if (not os.path.exists(HOSTCHECK_LIB)) or any(os.path.getmtime(d) > os.path.getmtime(HOSTCHECK_LIB) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                               "-std=c++17"] + (["-DCLOUDSC2_SINGLE"] if B.SINGLE else []) + ["-o", HOSTCHECK_LIB, src])
    return HOSTCHECK_LIB


_hc = None


def hostcheck():
    """The per-column device functions compiled for the host (tests/hostcheck) -- a unit-test vehicle, not a product path."""
    global _hc
    if _hc is None:
        lib = C.CDLL(build_hostcheck())
        pp = C.POINTER(B.Params)
        lib.hostcheck_satur.argtypes = [pp, C.c_int, C.c_int, C.c_int, B.Field, B.Field, B.Field]
        lib.hostcheck_nl.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(B.Inputs), C.POINTER(B.Outputs),
                                     B.Field, C.c_double]
        lib.hostcheck_tl.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(B.Inputs), C.POINTER(B.Outputs),
                                     C.POINTER(B.Inputs), C.POINTER(B.Outputs)]
        lib.hostcheck_ad.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(B.Inputs), C.POINTER(B.Outputs),
                                     C.POINTER(B.Inputs), C.POINTER(B.Outputs), C.c_void_p]
        lib.hostcheck_set_self_increment.argtypes = [C.c_double]
        lib.hostcheck_set_yy.argtypes = [C.c_void_p]
        lib.hostcheck_set_ad_norms.argtypes = [C.c_void_p]
        lib.hostcheck_get_norm3_max.restype = C.c_double
        lib.hostcheck_taylor_sweep.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(B.Inputs), C.POINTER(B.Outputs),
                                               C.POINTER(B.Outputs), C.c_void_p]
        _hc = lib
    return _hc


def hfld(a: np.ndarray, offset: int = 0, stride: int | None = None) -> B.Field:
    assert a.dtype == B.REAL and a.flags["C_CONTIGUOUS"]
    f = B.Field()
    f.ptr = a.ctypes.data + B.REAL_BYTES * offset
    f.block_stride = stride if stride is not None else int(np.prod(a.shape[1:]))
    return f


def host_traj_blocks(st: c2.Cloudsc2State, qsat: np.ndarray | None = None):
    """Inputs/Outputs blocks with HOST pointers into a Cloudsc2State (same mapping as DeviceState)."""
    S = st.nproma * st.nlev
    i = B.Inputs()
    i.paph = hfld(st.PAPH); i.pap = hfld(st.PAP); i.q = hfld(st.PQ)
    i.qsat = hfld(qsat) if qsat is not None else B.Field()
    i.t = hfld(st.PT)
    i.l = hfld(st.PCLV, 0 * S, 5 * S); i.i = hfld(st.PCLV, 1 * S, 5 * S)
    i.lude = hfld(st.PLUDE); i.lu = hfld(st.PLU); i.mfu = hfld(st.PMFU); i.mfd = hfld(st.PMFD)
    i.gtent = hfld(st.B_CML, 0 * S, 8 * S); i.gtenq = hfld(st.B_CML, 2 * S, 8 * S)
    i.gtenl = hfld(st.B_CML, 3 * S, 8 * S); i.gteni = hfld(st.B_CML, 4 * S, 8 * S)
    i.supsat = hfld(st.PSUPSAT)
    o = B.Outputs()
    o.tent = hfld(st.B_LOC, 0 * S, 8 * S); o.tenq = hfld(st.B_LOC, 2 * S, 8 * S)
    o.tenl = hfld(st.B_LOC, 3 * S, 8 * S); o.teni = hfld(st.B_LOC, 4 * S, 8 * S)
    o.clc = hfld(st.PA); o.covptot = hfld(st.PCOVPTOT)
    o.fplsl = hfld(st.PFPLSL); o.fplsn = hfld(st.PFPLSN); o.fhpsl = hfld(st.PFHPSL); o.fhpsn = hfld(st.PFHPSN)
    return i, o


def flat_fields(kind: str, nb: int, nlev: int, nproma: int, fill: float = 0.0) -> dict:
    names = B.IN_NAMES if kind == "in" else B.OUT_NAMES
    return {n: np.full((nb, nlev + (1 if n in refcall.HALF else 0), nproma), fill, dtype=B.REAL) for n in names}


def flat_block(kind: str, arrays: dict):
    blk = B.Inputs() if kind == "in" else B.Outputs()
    for n, a in arrays.items():
        setattr(blk, n, hfld(a))
    return blk


def increments_of(st: c2.Cloudsc2State, qsat: np.ndarray, zero_supsat: bool = False) -> dict:
    """dx = 0.01*x for the 16 inputs, flat (NBLOCKS, NLEVx, NPROMA) (cloudsc_driver_tl_mod.F90:156-171)."""
    src = {"paph": st.PAPH, "pap": st.PAP, "q": st.PQ, "qsat": qsat, "t": st.PT, "l": st.PCLV[:, 0], "i": st.PCLV[:, 1],
           "lude": st.PLUDE, "lu": st.PLU, "mfu": st.PMFU, "mfd": st.PMFD, "gtent": st.B_CML[:, 0], "gtenq": st.B_CML[:, 2],
           "gtenl": st.B_CML[:, 3], "gteni": st.B_CML[:, 4], "supsat": st.PSUPSAT}
    out = {n: np.ascontiguousarray(a * B.REAL(0.01)) for n, a in src.items()}
    if zero_supsat:
        out["supsat"][...] = 0.0
    return out


def make_params(tab: dict, **flags) -> B.Params:
    return c2.default_params(c2.ceta_from_table(tab), **flags)


def set_lib_params(lib, prm: B.Params):
    lib.set_params(prm.doubles30(), prm.ceta_array(), lphylin=bool(prm.lphylin), levapls2=bool(prm.levapls2),
                   lregcl=bool(prm.lregcl))


def relerr(ref: np.ndarray, got: np.ndarray) -> float:
    """max |got-ref| / max|ref| (0 if both vanish)."""
    d = float(np.max(np.abs(got - ref))) if ref.size else 0.0
    m = float(np.max(np.abs(ref))) if ref.size else 0.0
    if m == 0.0:
        return 0.0 if d == 0.0 else np.inf
    return d / m
