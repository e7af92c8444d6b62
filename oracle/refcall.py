"""TEST INFRASTRUCTURE ONLY -- ctypes access to the two CPU checkers:

* ``RefLib``    : oracle/_ref/libcloudsc2_ref.so, the UNMODIFIED reference Fortran compiled by oracle/Makefile
                  (exists only where /root/reference was present at build time, or where the prebuilt .so travelled);
                  ``RefLib(single=True)`` is the same sources built with the reference's -DSINGLE (fp32 arrays);
* ``OracleLib`` : oracle/libcloudsc2_oracle.so, the plain-C restatement (oracle/cloudsc2_oracle.c).

Both take one NPROMA block at a time as Fortran ``(KLON, KLEV)`` arrays = numpy C-ordered ``(KLEV, KLON)``.
Nothing in the product path (package, C ABI) imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PATH = os.path.join(HERE, "_ref", "libcloudsc2_ref.so")
REF_PATH_SP = os.path.join(HERE, "_ref", "libcloudsc2_ref_sp.so")
ORACLE_PATH = os.path.join(HERE, "libcloudsc2_oracle.so")

_dp = C.POINTER(C.c_double)

IN16 = ("paph", "pap", "q", "qsat", "t", "l", "i", "lude", "lu", "mfu", "mfd", "gtent", "gtenq", "gtenl", "gteni", "supsat")
OUT10 = ("tent", "tenq", "tenl", "teni", "clc", "fplsl", "fplsn", "fhpsl", "fhpsn", "covptot")
HALF = {"paph", "fplsl", "fplsn", "fhpsl", "fhpsn"}


def _p(a: np.ndarray, dtype=np.float64):
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], f"need C-contiguous {np.dtype(dtype).name}"
    return a.ctypes.data_as(C.c_void_p)


def big_stack(fn, *args, stack_mb: int = 1024, **kw):
    """Run fn in a thread with a large stack: the reference AD kernel keeps 117 automatic (KLON,KLEV) arrays
    (cloudsc2ad.F90:228-292) and the reference's own env scripts use `ulimit -s unlimited`."""
    out = {}

    def run():
        try:
            out["v"] = fn(*args, **kw)
        except BaseException as e:  # noqa: BLE001
            out["e"] = e

    old = threading.stack_size(stack_mb * 1024 * 1024)
    try:
        th = threading.Thread(target=run)
        th.start()
        th.join()
    finally:
        threading.stack_size(old)
    if "e" in out:
        raise out["e"]
    return out.get("v")


def kernel_arg_order(inp: dict, out: dict):
    """Dummy-argument order of CLOUDSC2 (cloudsc2.F90:13-18) after the scalars."""
    return [inp["paph"], inp["pap"], inp["q"], inp["qsat"], inp["t"], inp["l"], inp["i"], inp["lude"], inp["lu"],
            inp["mfu"], inp["mfd"], out["tent"], inp["gtent"], out["tenq"], inp["gtenq"], out["tenl"], inp["gtenl"],
            out["teni"], inp["gteni"], inp["supsat"], out["clc"], out["fplsl"], out["fplsn"], out["fhpsl"], out["fhpsn"],
            out["covptot"]]


def new_outputs(klev: int, klon: int, fill: float = 0.0, dtype=np.float64) -> dict:
    return {n: np.full((klev + (1 if n in HALF else 0), klon), fill, dtype=dtype) for n in OUT10}


def new_inputs(klev: int, klon: int, fill: float = 0.0, dtype=np.float64) -> dict:
    return {n: np.full((klev + (1 if n in HALF else 0), klon), fill, dtype=dtype) for n in IN16}


class _KernelLib:
    """Common calling convention of the reference shim and the C oracle."""

    prefix = ""

    def __init__(self, path: str, dtype=np.float64):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.path = path
        self.dtype = dtype  # element type of the field arrays (constants and PTSPHY are always C doubles)
        vp = C.c_void_p
        f = getattr(self.lib, self.prefix + "set_params")
        f.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        f.restype = None
        f = getattr(self.lib, self.prefix + "satur")
        f.argtypes = [C.c_int] * 4 + [vp] * 3
        f.restype = None
        f = getattr(self.lib, self.prefix + "cloudsc2")
        f.argtypes = [C.c_int] * 5 + [C.c_double] + [vp] * 26
        f.restype = None
        for n in ("cloudsc2tl", "cloudsc2ad"):
            f = getattr(self.lib, self.prefix + n)
            f.argtypes = [C.c_int] * 5 + [C.c_double] + [vp] * 52
            f.restype = None

    def _a(self, a: np.ndarray):
        return _p(a, self.dtype)

    def set_params(self, doubles30: np.ndarray, ceta: np.ndarray, lphylin=True, levapls2=False, lregcl=False):
        r = np.ascontiguousarray(doubles30, dtype=np.float64)
        ce = np.ascontiguousarray(ceta, dtype=np.float64)
        assert r.size == 30
        getattr(self.lib, self.prefix + "set_params")(r.ctypes.data_as(_dp), int(lphylin), int(levapls2), int(lregcl),
                                                      int(ce.size), ce.ctypes.data_as(_dp))
        self.nlev = int(ce.size)

    def satur(self, pap: np.ndarray, t: np.ndarray, kfdia: int | None = None) -> np.ndarray:
        klev, klon = pap.shape
        q = np.zeros_like(pap)
        getattr(self.lib, self.prefix + "satur")(1, kfdia or klon, klon, klev, self._a(pap), self._a(t), self._a(q))
        return q

    def cloudsc2(self, ptsphy: float, inp: dict, out: dict | None = None, ldrain1d=False, kfdia: int | None = None) -> dict:
        klev, klon = inp["pap"].shape
        out = out if out is not None else new_outputs(klev, klon, dtype=self.dtype)
        args = kernel_arg_order(inp, out)
        # the kernels keep (KLON,KLEV) work arrays on the stack (cloudsc2.F90:176-190): large KLON needs a large stack
        big_stack(getattr(self.lib, self.prefix + "cloudsc2"), 1, kfdia or klon, klon, klev, int(ldrain1d), float(ptsphy),
                  *[self._a(a) for a in args])
        return out

    def cloudsc2tl(self, ptsphy: float, inp5: dict, dinp: dict, out5: dict | None = None, dout: dict | None = None,
                   ldrain1d=False, kfdia: int | None = None):
        klev, klon = inp5["pap"].shape
        out5 = out5 if out5 is not None else new_outputs(klev, klon, dtype=self.dtype)
        dout = dout if dout is not None else new_outputs(klev, klon, dtype=self.dtype)
        args = kernel_arg_order(inp5, out5) + kernel_arg_order(dinp, dout)
        big_stack(getattr(self.lib, self.prefix + "cloudsc2tl"), 1, kfdia or klon, klon, klev, int(ldrain1d), float(ptsphy),
                  *[self._a(a) for a in args])
        return out5, dout

    def cloudsc2ad(self, ptsphy: float, inp5: dict, ainp: dict, aout: dict, out5: dict | None = None, ldrain1d=False,
                   kfdia: int | None = None):
        """ainp (input adjoints, accumulated) and aout (output adjoints, zeroed on return) are modified in place."""
        klev, klon = inp5["pap"].shape
        out5 = out5 if out5 is not None else new_outputs(klev, klon, dtype=self.dtype)
        args = kernel_arg_order(inp5, out5) + kernel_arg_order(ainp, aout)
        fn = getattr(self.lib, self.prefix + "cloudsc2ad")
        big_stack(fn, 1, kfdia or klon, klon, klev, int(ldrain1d), float(ptsphy), *[self._a(a) for a in args])
        return out5


class RefLib(_KernelLib):
    prefix = "ref_"

    def __init__(self, path: str | None = None, single: bool = False):
        super().__init__(path or (REF_PATH_SP if single else REF_PATH), np.float32 if single else np.float64)
        self.lib.ref_driver.argtypes = [C.c_int] * 5 + [C.c_double] + [C.c_void_p] * 18
        self.lib.ref_driver.restype = None

    def driver(self, which: int, numomp: int, nproma: int, nlev: int, ngptot: int, ptsphy: float, arrays18):
        """CLOUDSC_DRIVER (0) / _TL (1) / _AD (2) on GLOBAL_STATE arrays in the order of cloudsc_driver_mod.F90:22-30.
        The TL and AD drivers print their verdict on stdout."""
        os.environ.setdefault("OMP_STACKSIZE", "1G")
        big_stack(self.lib.ref_driver, which, numomp, nproma, nlev, ngptot, float(ptsphy), *[self._a(a) for a in arrays18])


    # ---- the data formats either side of the path, straight from the reference's modules ----
    @staticmethod
    def _capture(fd: int, fn, *args):
        """Run fn(*args) with file descriptor fd (1 = Fortran unit 6, 2 = unit 0) redirected to a file; return what was written."""
        import tempfile

        with tempfile.TemporaryFile() as tmp:
            saved = os.dup(fd)
            try:
                os.dup2(tmp.fileno(), fd)
                fn(*args)
            finally:
                os.dup2(saved, fd)
                os.close(saved)
            tmp.seek(0)
            return tmp.read().decode()

    def get_offsets(self, irank: int, numproc: int, nlon: int, ngptot: int, ngptotg: int | None):
        """GET_OFFSETS (expand_mod.F90:30-46): (start, end, size), 1-based like the reference."""
        st, en, sz = C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_get_offsets.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int)] * 3
        self.lib.ref_get_offsets.restype = None
        self.lib.ref_get_offsets(irank, numproc, nlon, ngptot, -1 if ngptotg is None else ngptotg, C.byref(st), C.byref(en), C.byref(sz))
        return st.value, en.value, sz.value

    def expand(self, buffer: np.ndarray, nproma: int, ngptot: int) -> np.ndarray:
        """EXPAND_R2 / EXPAND_R3 (expand_mod.F90:270-335).  buffer: (NLEV, NLON) or (NDIM, NLEV, NLON) doubles = Fortran
        (NLON, NLEV[, NDIM]); returns (NBLOCKS, [NDIM,] NLEV, NPROMA) = Fortran (NPROMA, NLEV, [NDIM,] NBLOCKS)."""
        buf = np.ascontiguousarray(buffer, dtype=np.float64)
        nblocks = (ngptot + nproma - 1) // nproma
        nlon = buf.shape[-1]
        if buf.ndim == 2:
            field = np.full((nblocks, buf.shape[0], nproma), np.nan, dtype=self.dtype)
            self.lib.ref_expand_r2.argtypes = [C.c_void_p] * 2 + [C.c_int] * 5
            self.lib.ref_expand_r2.restype = None
            self.lib.ref_expand_r2(_p(buf), self._a(field), nlon, nproma, buf.shape[0], ngptot, nblocks)
        else:
            field = np.full((nblocks, buf.shape[0], buf.shape[1], nproma), np.nan, dtype=self.dtype)
            self.lib.ref_expand_r3.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6
            self.lib.ref_expand_r3.restype = None
            self.lib.ref_expand_r3(_p(buf), self._a(field), nlon, nproma, buf.shape[1], buf.shape[0], ngptot, nblocks)
        return field

    def validate(self, name: str, ref: np.ndarray, field: np.ndarray, ngptot: int, ngptotg: int | None = None) -> str:
        """VALIDATE_R2 / VALIDATE_R3 (validate_mod.F90:165-261) on blocked arrays (NBLOCKS, [NDIM,] NLEV, NPROMA): the line
        ERROR_PRINT writes (validate_mod.F90:263-296), without the newline."""
        r = np.ascontiguousarray(ref, dtype=self.dtype)
        f = np.ascontiguousarray(field, dtype=self.dtype)
        assert r.shape == f.shape
        nm = name.encode()
        g = -1 if ngptotg is None else ngptotg
        if r.ndim == 3:
            nb, nlev, nproma = r.shape
            self.lib.ref_validate_r2.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 5
            self.lib.ref_validate_r2.restype = None
            out = self._capture(1, self.lib.ref_validate_r2, nm, len(nm), self._a(r), self._a(f), nproma, nlev, ngptot, nb, g)
        else:
            nb, ndim, nlev, nproma = r.shape
            self.lib.ref_validate_r3.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 6
            self.lib.ref_validate_r3.restype = None
            out = self._capture(1, self.lib.ref_validate_r3, nm, len(nm), self._a(r), self._a(f), nproma, nlev, ndim, ngptot, nb, g)
        return out.rstrip("\n")

    def error_print(self, name: str, zminval, zmaxval, zmaxerr, zerrsum, zsum, zavgpgp, ndim: int) -> str:
        """ERROR_PRINT (validate_mod.F90:263-296) for given statistics."""
        nm = name.encode()
        self.lib.ref_error_print.argtypes = [C.c_char_p, C.c_int] + [C.c_double] * 6 + [C.c_int]
        self.lib.ref_error_print.restype = None
        return self._capture(1, self.lib.ref_error_print, nm, len(nm), float(zminval), float(zmaxval), float(zmaxerr), float(zerrsum),
                             float(zsum), float(zavgpgp), ndim).rstrip("\n")

    def timer_print(self, nproma: int, ngpblks: int, ngptot: int, tthread, coreid, icalls, igpc, tdiff: float,
                    zhpm: float = 3996006.0) -> str:
        """PERFORMANCE_TIMER%PRINT_PERFORMANCE (timer_mod.F90:114-174) for given per-thread seconds / core ids / calls / columns
        and region seconds: the table it writes on unit 0."""
        tt = np.ascontiguousarray(tthread, dtype=np.float64)
        ints = [np.ascontiguousarray(a, dtype=np.int32) for a in (coreid, icalls, igpc)]
        self.lib.ref_timer_print.argtypes = [C.c_int] * 4 + [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        self.lib.ref_timer_print.restype = None
        return self._capture(2, self.lib.ref_timer_print, int(tt.size), nproma, ngpblks, ngptot, float(zhpm), _p(tt),
                             *[_p(a, np.int32) for a in ints], float(tdiff))


class OracleLib(_KernelLib):
    prefix = "oracle_"

    def __init__(self, path: str = ORACLE_PATH):
        super().__init__(path)


def have_ref(single: bool = False) -> bool:
    return os.path.exists(REF_PATH_SP if single else REF_PATH)


def have_oracle() -> bool:
    return os.path.exists(ORACLE_PATH)


# ---------------------------------------------------------------------------------------------------------------------
# block views of a Cloudsc2State
# ---------------------------------------------------------------------------------------------------------------------
def block_inputs(st, ibl: int, qsat: np.ndarray | None = None) -> dict:
    """The 16 kernel inputs of block ibl as contiguous (KLEV[+1], KLON) arrays (cloudsc_driver_mod.F90:94-107)."""
    c = np.ascontiguousarray
    return {
        "paph": c(st.PAPH[ibl]), "pap": c(st.PAP[ibl]), "q": c(st.PQ[ibl]),
        "qsat": c(qsat) if qsat is not None else np.zeros_like(st.PAP[ibl]), "t": c(st.PT[ibl]),
        "l": c(st.PCLV[ibl, 0]), "i": c(st.PCLV[ibl, 1]), "lude": c(st.PLUDE[ibl]), "lu": c(st.PLU[ibl]),
        "mfu": c(st.PMFU[ibl]), "mfd": c(st.PMFD[ibl]), "gtent": c(st.B_CML[ibl, 0]), "gtenq": c(st.B_CML[ibl, 2]),
        "gtenl": c(st.B_CML[ibl, 3]), "gteni": c(st.B_CML[ibl, 4]), "supsat": c(st.PSUPSAT[ibl]),
    }


def state_outputs_block(st, ibl: int) -> dict:
    return {"tent": st.B_LOC[ibl, 0], "tenq": st.B_LOC[ibl, 2], "tenl": st.B_LOC[ibl, 3], "teni": st.B_LOC[ibl, 4],
            "clc": st.PA[ibl], "fplsl": st.PFPLSL[ibl], "fplsn": st.PFPLSN[ibl], "fhpsl": st.PFHPSL[ibl],
            "fhpsn": st.PFHPSN[ibl], "covptot": st.PCOVPTOT[ibl]}
