"""ctypes binding of the C ABI declared in include/cloudsc2_hip.h (libcloudsc2_hip.so, built in-tree by
``__graft_entry__.build()`` / ``csrc/Makefile``).

There is deliberately no fallback: if the HIP library is missing, importing this module raises, and every compute
entry point of the library itself returns CLOUDSC2_ENODEVICE when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

CLOUDSC2_MAX_NLEV = 200
CLOUDSC2_EINVAL, CLOUDSC2_ENODEVICE, CLOUDSC2_ETLWRONG = -1, -2, -3

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLOUDSC2_PRECISION=single: the process works on fp32 arrays through libcloudsc2_hip_sp.so -- the counterpart of
# building the reference with -DSINGLE (JPRB = JPRM, src/common/module/parkind1.F90:40-41).  Like there, it is a
# whole-program choice: one process, one precision.
SINGLE = os.environ.get("CLOUDSC2_PRECISION", "double").strip().lower() in ("single", "sp", "f32", "fp32", "float32")
REAL = np.float32 if SINGLE else np.float64
REAL_BYTES = 4 if SINGLE else 8
c_real = C.c_float if SINGLE else C.c_double
# CLOUDSC2_LIB: load another build of the same library (kernel-tuning experiments)
LIB_PATH = os.environ.get("CLOUDSC2_LIB") or os.path.join(_HERE, "csrc", "libcloudsc2_hip_sp.so" if SINGLE else "libcloudsc2_hip.so")


def torch_real():
    import torch

    return torch.float32 if SINGLE else torch.float64


class Cloudsc2Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cloudsc2 error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """struct cloudsc2_params"""

    _DOUBLES = ("rg", "rd", "rcpd", "retv", "rlvtt", "rlstt", "rlmlt", "rtt", "r2es", "r3les", "r3ies", "r4les",
                "r4ies", "r5les", "r5ies", "r5alvcp", "r5alscp", "ralvdcp", "ralsdcp", "rtwat", "rtice",
                "rtwat_rtice_r", "rvtmp2", "rclcrit", "rkconv", "rlmin", "rpecons", "rlptrc", "rticecu",
                "rtwat_rticecu_r")
    _fields_ = ([(n, C.c_double) for n in _DOUBLES] +
                [("lphylin", C.c_int), ("levapls2", C.c_int), ("lregcl", C.c_int), ("ldrain1d", C.c_int),
                 ("nlev", C.c_int), ("math_mode", C.c_int), ("ceta", C.c_double * CLOUDSC2_MAX_NLEV)])

    def doubles30(self) -> np.ndarray:
        """The 30 leading constants in the order oracle/ref_harness.F90 takes them."""
        return np.array([getattr(self, n) for n in self._DOUBLES], dtype=np.float64)

    def set_ceta(self, ceta) -> "Params":
        ceta = np.asarray(ceta, dtype=np.float64)
        if ceta.size > CLOUDSC2_MAX_NLEV:
            raise ValueError("nlev exceeds CLOUDSC2_MAX_NLEV")
        self.nlev = int(ceta.size)
        for k, v in enumerate(ceta):
            self.ceta[k] = float(v)
        return self

    def ceta_array(self) -> np.ndarray:
        return np.array(self.ceta[: self.nlev], dtype=np.float64)


class Field(C.Structure):
    """struct cloudsc2_field"""

    _fields_ = [("ptr", C.c_void_p), ("block_stride", C.c_longlong)]


IN_NAMES = ("paph", "pap", "q", "qsat", "t", "l", "i", "lude", "lu", "mfu", "mfd", "gtent", "gtenq", "gtenl",
            "gteni", "supsat")
OUT_NAMES = ("tent", "tenq", "tenl", "teni", "clc", "fplsl", "fplsn", "fhpsl", "fhpsn", "covptot")


class Inputs(C.Structure):
    """struct cloudsc2_inputs"""

    _fields_ = [(n, Field) for n in IN_NAMES]


class Outputs(C.Structure):
    """struct cloudsc2_outputs"""

    _fields_ = [(n, Field) for n in OUT_NAMES]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C dwarf_p_cloudsc2_tl_ad_amd/csrc`). There is no CPU fallback.")
    # One HIP/HSA runtime per process: the torch wheel bundles its own libamdhip64.so.7 / libhsa-runtime64 and a second
    # copy (the system ROCm one this library was linked against) cannot open the GPU any more.  Loading torch first
    # makes the dynamic loader resolve this library's NEEDED libamdhip64.so.7 to the copy that is already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    rp = C.POINTER(c_real)  # cloudsc2_real*
    pp = C.POINTER(Params)
    lib.cloudsc2_real_bytes.restype = C.c_int
    if lib.cloudsc2_real_bytes() != REAL_BYTES:
        raise ImportError(f"{LIB_PATH} works on {lib.cloudsc2_real_bytes()}-byte reals, CLOUDSC2_PRECISION asks for {REAL_BYTES}")
    lib.cloudsc2_params_default.argtypes = [pp]
    lib.cloudsc2_params_default.restype = None
    lib.cloudsc2_last_error.restype = C.c_char_p
    lib.cloudsc2_device_available.restype = C.c_int
    lib.cloudsc2_set_math_mode.argtypes = [C.c_int]
    lib.cloudsc2_set_math_mode.restype = None
    lib.cloudsc2_get_math_mode.restype = C.c_int
    lib.cloudsc2_nl_launch.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                       Field, C.c_double, C.c_void_p]
    lib.cloudsc2_satur_launch.argtypes = [pp, C.c_int, C.c_int, C.c_int, Field, Field, Field, C.c_void_p]
    lib.cloudsc2_tl_launch.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                       C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p]
    lib.cloudsc2_tl_launch_self.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs), C.c_double,
                                            C.POINTER(Outputs), C.c_void_p, C.c_void_p]
    lib.cloudsc2_ad_launch_reverse_norms.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                                     C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cloudsc2_ad_launch_reverse_norms.restype = C.c_int
    lib.cloudsc2_tl_launch_self.restype = C.c_int
    lib.cloudsc2_ad_launch.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                       C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p, C.c_void_p]
    lib.cloudsc2_ad_launch_assign.argtypes = lib.cloudsc2_ad_launch.argtypes
    lib.cloudsc2_ad_launch_assign.restype = C.c_int
    lib.cloudsc2_ad_launch_forward.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                               C.c_void_p, C.c_void_p]
    lib.cloudsc2_ad_launch_forward.restype = C.c_int
    lib.cloudsc2_ad_launch_reverse.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                               C.POINTER(Inputs), C.POINTER(Outputs), C.c_void_p, C.c_int, C.c_void_p]
    lib.cloudsc2_ad_launch_reverse.restype = C.c_int
    lib.cloudsc2_taylor_sums_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Outputs), C.POINTER(Outputs),
                                                C.POINTER(Outputs), C.c_double, C.c_void_p, C.c_void_p]
    lib.cloudsc2_taylor_sweep_work_doubles.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    lib.cloudsc2_taylor_sweep_work_doubles.restype = C.c_int
    lib.cloudsc2_taylor_sweep_launch.argtypes = [pp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Outputs),
                                                 C.POINTER(Outputs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cloudsc2_taylor_sweep_launch.restype = C.c_int
    lib.cloudsc2_adjoint_norms_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Inputs), C.POINTER(Field),
                                                  C.POINTER(Outputs), C.POINTER(Inputs), C.c_void_p, C.c_void_p,
                                                  C.c_void_p]
    host18 = [rp] * 18
    lib.cloudsc2_nl_run.argtypes = [pp, C.c_int, C.c_int, C.c_int, C.c_double] + host18 + [dp]
    lib.cloudsc2_tl_taylor_run.argtypes = [pp, C.c_int, C.c_int, C.c_int, C.c_double] + host18 + [dp, dp]
    lib.cloudsc2_ad_symmetry_run.argtypes = [pp, C.c_int, C.c_int, C.c_int, C.c_double] + host18 + [dp, dp]
    lib.cloudsc2_release_workspace.restype = None
    lib.cloudsc2_state_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.cloudsc2_state_destroy.argtypes = [C.c_void_p]
    lib.cloudsc2_state_destroy.restype = None
    lib.cloudsc2_state_field.argtypes = [C.c_void_p, C.c_int, C.POINTER(Field)]
    lib.cloudsc2_state_blocking.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cloudsc2_state_blocking.restype = C.c_int
    lib.cloudsc2_state_expand.argtypes = [C.c_void_p, C.c_int, rp, C.c_int, C.c_int, C.c_longlong]
    lib.cloudsc2_state_upload.argtypes = [C.c_void_p] + host18
    lib.cloudsc2_state_download.argtypes = [C.c_void_p] + [rp] * 7
    lib.cloudsc2_state_nl.argtypes = [C.c_void_p, pp, C.c_double, C.c_int, dp]
    lib.cloudsc2_state_tl_taylor.argtypes = [C.c_void_p, pp, C.c_double, dp, dp]
    lib.cloudsc2_state_ad_symmetry.argtypes = [C.c_void_p, pp, C.c_double, dp, dp]
    lib.cloudsc2_state_validate.argtypes = [C.c_void_p, C.c_int, C.c_int, rp, C.c_int, C.c_int, C.c_longlong, dp]
    for name in ("cloudsc2_state_create", "cloudsc2_state_field", "cloudsc2_state_expand", "cloudsc2_state_upload",
                 "cloudsc2_state_download", "cloudsc2_state_nl", "cloudsc2_state_tl_taylor", "cloudsc2_state_ad_symmetry",
                 "cloudsc2_state_validate"):
        getattr(lib, name).restype = C.c_int
    lib.cloudsc2_device_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.cloudsc2_device_free.argtypes = [C.c_void_p]
    lib.cloudsc2_device_malloc_info.argtypes = [C.POINTER(C.c_int), dp, dp, dp]
    lib.cloudsc2_device_malloc_info.restype = None
    lib.cloudsc2_device_malloc_counts.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.cloudsc2_device_malloc_counts.restype = None
    lib.cloudsc2_pace_plan.argtypes = [C.c_longlong, C.c_longlong, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cloudsc2_pace_plan.restype = C.c_int
    lib.cloudsc2_simd_population.argtypes = [C.c_longlong, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cloudsc2_simd_population.restype = C.c_int
    lib.cloudsc2_dispatch_probe.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.cloudsc2_dispatch_probe.restype = C.c_int
    lib.cloudsc2_pace_probe.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.cloudsc2_pace_probe.restype = C.c_int
    lib.cloudsc2_synthetic_table.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 12
    lib.cloudsc2_synthetic_table.restype = C.c_int
    lib.cloudsc2_device_prepare.argtypes = []
    lib.cloudsc2_device_prepare.restype = C.c_int
    lib.cloudsc2_device_rules.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cloudsc2_device_rules.restype = C.c_int
    lib.cloudsc2_kernel_occupancy.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.cloudsc2_kernel_occupancy.restype = C.c_int
    lib.cloudsc2_device_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, dp]
    lib.cloudsc2_device_probe.restype = C.c_int
    lib.cloudsc2_device_malloc_state.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.cloudsc2_device_malloc_state.restype = C.c_int
    lib.cloudsc2_taylor_verdict.argtypes = [dp, C.POINTER(C.c_int)]
    lib.cloudsc2_adjoint_verdict.argtypes = [C.c_double]
    expand_args = [rp, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_longlong, Field]
    lib.cloudsc2_expand_launch.argtypes = expand_args + [C.c_void_p]
    lib.cloudsc2_validate_workspace_doubles.restype = C.c_int
    lib.cloudsc2_validate_launch.argtypes = expand_args + [dp, dp, C.c_void_p]
    lib.cloudsc2_expand_offsets.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong),
                                            C.POINTER(C.c_int)]
    lib.cloudsc2_expand_offsets.restype = None
    lib.cloudsc2_validate_relerr.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cloudsc2_validate_relerr.restype = C.c_double
    lib.cloudsc2_validate_format.argtypes = [C.c_char_p, C.c_int, dp, C.c_longlong, C.c_char_p, C.c_int]
    lib.cloudsc2_validate_header.argtypes = [C.c_char_p, C.c_int]
    for name in ("cloudsc2_expand_launch", "cloudsc2_validate_launch", "cloudsc2_validate_format", "cloudsc2_validate_header",
                 "cloudsc2_nl_launch", "cloudsc2_satur_launch", "cloudsc2_tl_launch", "cloudsc2_ad_launch",
                 "cloudsc2_taylor_sums_launch", "cloudsc2_adjoint_norms_launch", "cloudsc2_nl_run",
                 "cloudsc2_tl_taylor_run", "cloudsc2_ad_symmetry_run", "cloudsc2_taylor_verdict",
                 "cloudsc2_adjoint_verdict", "cloudsc2_device_malloc", "cloudsc2_device_free"):
        getattr(lib, name).restype = C.c_int
    return lib


lib = _load()

# every symbol include/cloudsc2_hip.h declares
EXPORTED = ("cloudsc2_params_default", "cloudsc2_last_error", "cloudsc2_device_available", "cloudsc2_current_device", "cloudsc2_set_math_mode",
            "cloudsc2_get_math_mode", "cloudsc2_real_bytes", "cloudsc2_nl_launch",
            "cloudsc2_satur_launch", "cloudsc2_tl_launch", "cloudsc2_tl_launch_self", "cloudsc2_ad_launch", "cloudsc2_ad_launch_assign",
            "cloudsc2_ad_launch_forward", "cloudsc2_ad_launch_reverse", "cloudsc2_ad_launch_reverse_norms", "cloudsc2_taylor_sums_launch",
            "cloudsc2_taylor_sweep_work_doubles", "cloudsc2_taylor_sweep_launch", "cloudsc2_adjoint_norms_launch", "cloudsc2_nl_run", "cloudsc2_tl_taylor_run", "cloudsc2_ad_symmetry_run",
            "cloudsc2_release_workspace", "cloudsc2_taylor_verdict", "cloudsc2_adjoint_verdict",
            "cloudsc2_expand_launch", "cloudsc2_validate_workspace_doubles", "cloudsc2_validate_launch",
            "cloudsc2_expand_offsets", "cloudsc2_validate_relerr", "cloudsc2_validate_format", "cloudsc2_validate_header",
            "cloudsc2_device_malloc", "cloudsc2_device_free", "cloudsc2_device_malloc_info", "cloudsc2_device_malloc_counts", "cloudsc2_pace_plan", "cloudsc2_simd_population", "cloudsc2_dispatch_probe", "cloudsc2_pace_probe", "cloudsc2_device_prepare", "cloudsc2_device_rules", "cloudsc2_kernel_occupancy", "cloudsc2_device_probe", "cloudsc2_device_malloc_state",
            "cloudsc2_state_create", "cloudsc2_state_destroy", "cloudsc2_state_field", "cloudsc2_state_blocking", "cloudsc2_state_expand",
            "cloudsc2_state_upload", "cloudsc2_state_download", "cloudsc2_state_nl", "cloudsc2_state_tl_taylor",
            "cloudsc2_state_ad_symmetry", "cloudsc2_state_validate", "cloudsc2_synthetic_table")

# field ids of the resident-state API (enum in include/cloudsc2_hip.h)
F_FULL = {"PT": 0, "PQ": 1, "PAP": 2, "PAPH": 3, "PLU": 4, "PLUDE": 5, "PMFU": 6, "PMFD": 7, "PA": 8, "PSUPSAT": 9, "PCOVPTOT": 10,
          "PFPLSL": 11, "PFPLSN": 12, "PFHPSL": 13, "PFHPSN": 14, "QSAT": 15}
F_CML_T, F_LOC_T, F_PCLV_QL = 16, 24, 32


def check(rc: int) -> None:
    if rc != 0:
        raise Cloudsc2Error(rc, (lib.cloudsc2_last_error() or b"").decode())


class DeviceBuffer:
    """Device memory from cloudsc2_device_malloc (placed: the fastest of several candidate allocations for the sweeps' write
    stream, include/cloudsc2_hip.h).  Exposes __cuda_array_interface__ so that torch can view it; freed with the last
    reference."""

    def __init__(self, nbytes: int, state_geom=None):
        p = C.c_void_p()
        if state_geom is None:
            check(lib.cloudsc2_device_malloc(C.byref(p), int(nbytes)))
        else:  # (nproma, nlev, ngptot): a state of this geometry sits at the start; placement judged by the NL sweep itself
            check(lib.cloudsc2_device_malloc_state(C.byref(p), int(nbytes), *[int(x) for x in state_geom]))
        self.ptr, self.nbytes = int(p.value or 0), int(nbytes)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        if getattr(self, "ptr", 0):
            lib.cloudsc2_device_free(C.c_void_p(self.ptr))
            self.ptr = 0


def device_probe(ptr: int, nbytes: int, kind: int = 0, rounds: int = 5) -> float:
    """Diagnostic: median ms of the allocator's probe stream `kind` (0 writes, 1 the NL sweep's pattern) over the buffer, which it
    overwrites (cloudsc2_device_probe)."""
    ms = C.c_double()
    check(lib.cloudsc2_device_probe(C.c_void_p(ptr), int(nbytes), int(kind), int(rounds), C.byref(ms)))
    return ms.value


def device_malloc_info() -> dict:
    """Placement report of the most recent cloudsc2_device_malloc of this process."""
    c = C.c_int()
    b, m, w = C.c_double(), C.c_double(), C.c_double()
    lib.cloudsc2_device_malloc_info(C.byref(c), C.byref(b), C.byref(m), C.byref(w))
    return {"candidates": c.value, "probe_ms_best": b.value, "probe_ms_median": m.value, "probe_ms_worst": w.value}


def device_malloc_counts() -> tuple:
    """(searched, plain): allocations of this process that went through the placement search / were one plain hipMalloc."""
    a, b = C.c_longlong(), C.c_longlong()
    lib.cloudsc2_device_malloc_counts(C.byref(a), C.byref(b))
    return a.value, b.value


# CLOUDSC2_STATE_ALLOC=torch: the Python mirror's device arrays come from torch's caching allocator (plain hipMalloc) instead
# of cloudsc2_device_malloc -- A/B measurements of the placement effect only.
STATE_ALLOC_TORCH = os.environ.get("CLOUDSC2_STATE_ALLOC", "library").strip().lower() == "torch"


class DeviceArena:
    """ONE placed allocation (cloudsc2_device_malloc) from which a set of arrays is carved, 256-byte aligned -- the Python
    counterpart of the library's own Arena (csrc/cloudsc2_driver.inc).  One allocation per state means one placement scan
    per state, and no array wastes the tail of a chunk."""

    ALIGN = 256

    def __init__(self, nbytes: int, device="cuda:0", state_geom=None):
        import torch

        self.device = torch.device(device)
        self.nbytes = int(nbytes)
        self.used = 0
        self.info = {}
        if STATE_ALLOC_TORCH:
            self.buf = None
            self.raw = torch.empty(max(self.nbytes, 1), dtype=torch.uint8, device=self.device)
        else:
            with torch.cuda.device(self.device):
                self.buf = DeviceBuffer(max(self.nbytes, 1), state_geom)
                self.raw = torch.as_tensor(self.buf, device=self.device)  # uint8 view; keeps `buf` alive
            self.info = device_malloc_info()

    @staticmethod
    def size_of(shapes, itemsize: int) -> int:
        total = 0
        for shp in shapes:
            n = itemsize
            for d in shp:
                n *= int(d)
            total += (n + DeviceArena.ALIGN - 1) // DeviceArena.ALIGN * DeviceArena.ALIGN
        return total

    def take(self, shape, dtype=None, zero: bool = False):
        import torch

        dtype = torch_real() if dtype is None else dtype
        itemsize = torch.empty((), dtype=dtype).element_size()
        n = itemsize
        for d in shape:
            n *= int(d)
        if self.used + n > self.nbytes:
            raise MemoryError("DeviceArena exhausted")
        t = self.raw[self.used: self.used + n].view(dtype).reshape(tuple(shape))
        self.used += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        t._cloudsc2_arena = self  # the view holds the owner
        return t.zero_() if zero else t


def device_empty(shape, dtype=None, device="cuda:0"):
    """An uninitialised torch tensor of `shape` in memory from cloudsc2_device_malloc (torch only VIEWS the memory)."""
    import torch

    dtype = torch_real() if dtype is None else dtype
    itemsize = torch.empty((), dtype=dtype).element_size()
    return DeviceArena(DeviceArena.size_of([shape], itemsize), device).take(shape, dtype)


def device_zeros(shape, dtype=None, device="cuda:0"):
    return device_empty(shape, dtype, device).zero_()


def default_params(ceta=None, *, lregcl: bool = False, levapls2: bool = False, ldrain1d: bool = False) -> Params:
    p = Params()
    lib.cloudsc2_params_default(C.byref(p))
    p.lregcl = int(lregcl)
    p.levapls2 = int(levapls2)
    p.ldrain1d = int(ldrain1d)
    if ceta is not None:
        p.set_ceta(ceta)
    return p


def set_math_mode(precise: bool) -> None:
    """False: fast math (default); True: IEEE division + libm exp in the reference's operation order."""
    lib.cloudsc2_set_math_mode(int(bool(precise)))


def get_math_mode() -> bool:
    return bool(lib.cloudsc2_get_math_mode())


def device_available() -> bool:
    return bool(lib.cloudsc2_device_available())


def taylor_verdict(znormg) -> tuple[bool, int]:
    z = (C.c_double * 10)(*[float(v) for v in znormg])
    itest = C.c_int(0)
    ok = lib.cloudsc2_taylor_verdict(z, C.byref(itest))
    return bool(ok), int(itest.value)


def adjoint_verdict(znormg: float) -> bool:
    return bool(lib.cloudsc2_adjoint_verdict(float(znormg)))


def expand_offsets(klon: int, ngptot: int, ngptotg: int = 0, irank: int = 0, numproc: int = 1):
    """GET_OFFSETS (expand_mod.F90:30-46): (0-based first table column, tiling period) of a rank."""
    start, period = C.c_longlong(), C.c_int()
    lib.cloudsc2_expand_offsets(klon, ngptot, ngptotg, irank, numproc, C.byref(start), C.byref(period))
    return start.value, period.value


def validate_line(name: str, ndim: int, stats, ngptotg: int) -> str:
    """The line ERROR_PRINT writes for one variable (validate_mod.F90:263-296)."""
    buf = C.create_string_buffer(200)
    st = (C.c_double * 5)(*[float(x) for x in stats])
    check(lib.cloudsc2_validate_format(name.encode(), ndim, st, int(ngptotg), buf, 200))
    return buf.value.decode()


def validate_header() -> str:
    buf = C.create_string_buffer(200)
    check(lib.cloudsc2_validate_header(buf, 200))
    return buf.value.decode()
