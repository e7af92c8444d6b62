"""input.h5 / reference.h5 of the dwarf through libcloudsc2_io.so (include/cloudsc2_io.h; HDF5 C API, no h5py).

Mirrors what the reference does in ``CLOUDSC2_ARRAY_STATE_LOAD`` (src/common/module/cloudsc2_array_state_mod.F90:153-203),
``..._WRITE_REFERENCE`` (:260-287) and the four ``*_LOAD_PARAMETERS`` routines.  A "table" is the dict of
``(NLEV[+1], KLON)`` arrays ``state.synthetic_table`` produces -- the same bytes as the file's ``(KLEV, KLON)`` datasets --
so ``state.state_from_table`` (host tiling) and ``DeviceState.from_table`` (device tiling) consume either source.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import binding as _b
from .state import PLANE_A, PLANE_Q, PLANE_QI, PLANE_QL, PLANE_T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBIO = None

INPUT_FIELDS_2D = ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT")  # :162-172
REFERENCE_FIELDS = ("PLUDE", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "TENDENCY_LOC_A", "TENDENCY_LOC_Q",
                    "TENDENCY_LOC_T", "TENDENCY_LOC_CLD")  # :275-284
NCLV = 5


def _lib():
    global _LIBIO
    if _LIBIO is None:
        path = os.environ.get("CLOUDSC2_IO_LIB", os.path.join(_HERE, "csrc", "libcloudsc2_io.so"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built (make -C dwarf_p_cloudsc2_tl_ad_amd/csrc libcloudsc2_io.so; needs hdf5.h)")
        lib = C.CDLL(path)
        lib.cloudsc2_io_last_error.restype = C.c_char_p
        lib.cloudsc2_file_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.cloudsc2_file_close.argtypes = [C.c_void_p]
        lib.cloudsc2_file_has.argtypes = [C.c_void_p, C.c_char_p]
        lib.cloudsc2_file_shape.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
        lib.cloudsc2_file_read_f64.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
        lib.cloudsc2_file_read_i32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
        lib.cloudsc2_file_write_f64.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.c_void_p]
        lib.cloudsc2_file_write_i32.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.c_void_p]
        lib.cloudsc2_file_read_params.argtypes = [C.c_void_p, C.POINTER(_b.Params), C.POINTER(C.c_double)]
        lib.cloudsc2_file_write_params.argtypes = [C.c_void_p, C.POINTER(_b.Params), C.c_double, C.c_int]
        _LIBIO = lib
    return _LIBIO


class IOError_(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        raise IOError_(f"{what}: rc={rc}: {_lib().cloudsc2_io_last_error().decode()}")


class H5File:
    """One input.h5 / reference.h5 style file.  mode 'r' or 'w' (create/truncate)."""

    def __init__(self, path: str, mode: str = "r"):
        self._h = C.c_void_p()
        _check(_lib().cloudsc2_file_open(os.fsencode(path), 1 if mode == "w" else 0, C.byref(self._h)), f"open {path}")

    def close(self):
        if self._h:
            _check(_lib().cloudsc2_file_close(self._h), "close")
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def has(self, name: str) -> bool:
        return bool(_lib().cloudsc2_file_has(self._h, name.encode()))

    def shape(self, name: str) -> tuple:
        nd = C.c_int()
        dims = (C.c_longlong * 4)()
        _check(_lib().cloudsc2_file_shape(self._h, name.encode(), C.byref(nd), dims), f"shape {name}")
        return tuple(int(dims[i]) for i in range(nd.value))

    def read(self, name: str, dtype=np.float64) -> np.ndarray:
        shp = self.shape(name)
        a = np.empty(shp, dtype=dtype)
        fn = _lib().cloudsc2_file_read_f64 if dtype == np.float64 else _lib().cloudsc2_file_read_i32
        _check(fn(self._h, name.encode(), a.ctypes.data, a.size), f"read {name}")
        return a

    def scalar(self, name: str, dtype=np.float64):
        return self.read(name, dtype).reshape(-1)[0].item()

    def write(self, name: str, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.float64:
            fn = _lib().cloudsc2_file_write_f64
        elif a.dtype == np.int32:
            fn = _lib().cloudsc2_file_write_i32
        else:
            raise TypeError(a.dtype)
        if a.ndim == 0:
            a = a.reshape(1)
        dims = (C.c_longlong * a.ndim)(*a.shape)
        _check(fn(self._h, name.encode(), a.ndim, dims, a.ctypes.data), f"write {name}")

    def read_params(self, lregcl: bool = False, ldrain1d: bool = False):
        """(Params, PTSPHY): the module constants as the reference loads them, CETA from column 1 (dwarf_cloudsc.F90:100-102)."""
        prm = _b.Params()
        prm.lregcl = int(lregcl)
        prm.ldrain1d = int(ldrain1d)
        pts = C.c_double()
        _check(_lib().cloudsc2_file_read_params(self._h, C.byref(prm), C.byref(pts)), "read_params")
        return prm, pts.value

    def write_params(self, prm, ptsphy: float, klon: int):
        _check(_lib().cloudsc2_file_write_params(self._h, C.byref(prm), float(ptsphy), int(klon)), "write_params")


def write_input_file(path: str, tab: dict, prm) -> None:
    """A table in the dataset layout CLOUDSC2_ARRAY_STATE_LOAD reads (cloudsc2_array_state_mod.F90:159-199)."""
    nlev, klon = tab["PT"].shape
    zeros = np.zeros((nlev, klon))
    with H5File(path, "w") as f:
        f.write_params(prm, float(tab["PTSPHY"]), klon)
        for n in INPUT_FIELDS_2D:
            f.write(n, np.asarray(tab[n], dtype=np.float64))
        clv = np.zeros((NCLV, nlev, klon))
        clv[0], clv[1] = tab["PCLV_QL"], tab["PCLV_QI"]
        f.write("PCLV", clv)
        f.write("TENDENCY_CML_T", np.asarray(tab["TENDENCY_CML_T"], dtype=np.float64))
        f.write("TENDENCY_CML_A", np.asarray(tab.get("TENDENCY_CML_A", zeros), dtype=np.float64))
        f.write("TENDENCY_CML_Q", np.asarray(tab["TENDENCY_CML_Q"], dtype=np.float64))
        cld = np.zeros((NCLV, nlev, klon))
        cld[0], cld[1] = tab["TENDENCY_CML_QL"], tab["TENDENCY_CML_QI"]
        f.write("TENDENCY_CML_CLD", cld)
        f.write("LDSLPHY", np.array([0], dtype=np.int32))
        f.write("LDMAINCALL", np.array([1], dtype=np.int32))


def read_input_file(path: str, lregcl: bool = False, ldrain1d: bool = False):
    """(table, Params): inverse of write_input_file; works on the reference's input.h5."""
    with H5File(path) as f:
        prm, ptsphy = f.read_params(lregcl=lregcl, ldrain1d=ldrain1d)
        tab = {n: f.read(n) for n in INPUT_FIELDS_2D}
        clv = f.read("PCLV")
        tab["PCLV_QL"], tab["PCLV_QI"] = np.ascontiguousarray(clv[0]), np.ascontiguousarray(clv[1])
        tab["TENDENCY_CML_T"] = f.read("TENDENCY_CML_T")
        tab["TENDENCY_CML_A"] = f.read("TENDENCY_CML_A")
        tab["TENDENCY_CML_Q"] = f.read("TENDENCY_CML_Q")
        cld = f.read("TENDENCY_CML_CLD")
        tab["TENDENCY_CML_QL"], tab["TENDENCY_CML_QI"] = np.ascontiguousarray(cld[0]), np.ascontiguousarray(cld[1])
        tab["PTSPHY"] = ptsphy
    return tab, prm


def reference_table_from_state(st, klon: int) -> dict:
    """The ten datasets WRITE_REFERENCE stores, from the first KLON columns of a computed state
    (cloudsc2_array_state_mod.F90:275-284; the reference insists on NPROMA = KLON = 100 and writes block 1)."""
    def cols(a):  # (NBLOCKS, NLEVx, NPROMA) -> (NLEVx, klon)
        nlevx = a.shape[1]
        return np.ascontiguousarray(a.transpose(1, 0, 2).reshape(nlevx, -1)[:, :klon])

    out = {"PLUDE": cols(st.PLUDE), "PCOVPTOT": cols(st.PCOVPTOT), "PFPLSL": cols(st.PFPLSL), "PFPLSN": cols(st.PFPLSN),
           "PFHPSL": cols(st.PFHPSL), "PFHPSN": cols(st.PFHPSN)}
    b = st.B_LOC  # (NBLOCKS, 8, NLEV, NPROMA)
    out["TENDENCY_LOC_A"] = cols(b[:, PLANE_A])
    out["TENDENCY_LOC_Q"] = cols(b[:, PLANE_Q])
    out["TENDENCY_LOC_T"] = cols(b[:, PLANE_T])
    out["TENDENCY_LOC_CLD"] = np.stack([cols(b[:, PLANE_QL + m]) for m in range(NCLV)])
    assert PLANE_QI == PLANE_QL + 1
    return out


def write_reference_file(path: str, ref: dict) -> None:
    nlev, klon = ref["PCOVPTOT"].shape
    with H5File(path, "w") as f:
        f.write("KLON", np.array([klon], dtype=np.int32))
        f.write("KLEV", np.array([nlev], dtype=np.int32))
        for n in REFERENCE_FIELDS:
            f.write(n, np.asarray(ref[n], dtype=np.float64))


def read_reference_file(path: str) -> dict:
    with H5File(path) as f:
        return {n: f.read(n) for n in REFERENCE_FIELDS}
