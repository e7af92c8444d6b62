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
