"""Host-side mirror of the reference's driver interface for the CLOUDSC2 hot path.

``cloudsc_driver``, ``cloudsc_driver_tl`` and ``cloudsc_driver_ad`` take the argument list of the reference's
``CLOUDSC_DRIVER{,_TL,_AD}`` (src/cloudsc2_nl/cloudsc_driver_mod.F90:22-30; the TL and AD drivers share it) on numpy
arrays in the GLOBAL_STATE layout and dispatch through the C ABI to the HIP kernels -- the same calls the Fortran
drivers in ``fortran/`` make through ISO_C_BINDING.  ``DeviceState`` keeps the whole state resident in HBM (torch
tensors are used only as device memory) and launches the kernels on a HIP stream; it is what ``bench.py`` times.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as B
from .state import PLANE_A, PLANE_Q, PLANE_QI, PLANE_QL, PLANE_QV, PLANE_T, Cloudsc2State, nblocks_of

_dp = C.POINTER(C.c_double)
_rp = C.POINTER(B.c_real)


def _ptr(a: np.ndarray):
    if a.dtype != B.REAL or not a.flags["C_CONTIGUOUS"]:
        raise ValueError(f"state arrays must be C-contiguous {np.dtype(B.REAL).name}")
    return a.ctypes.data_as(_rp)


def _host_args(prm: B.Params, nproma, nlev, ngptot, ptsphy, arrays):
    return [C.byref(prm), int(nproma), int(nlev), int(ngptot), float(ptsphy)] + [_ptr(a) for a in arrays]


def cloudsc_driver(prm: B.Params, numomp, nproma, nlev, ngptot, ngptotg, ptsphy, pt, pq, tendency_cml, tendency_loc, pap,
                   paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl, pfplsn, pfhpsl, pfhpsn) -> float:
    """CLOUDSC_DRIVER (cloudsc_driver_mod.F90:22-125).  ``tendency_cml`` / ``tendency_loc`` are the B_CML / B_LOC
    buffers the reference's STATE_TYPE arrays view.  ``numomp`` and ``ngptotg`` are accepted for signature parity
    (the block loop is one GPU launch).  Returns the kernel time in ms; outputs are written in place."""
    del numomp, ngptotg
    ms = C.c_double(0.0)
    arrays = [pt, pq, tendency_cml, tendency_loc, pap, paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl,
              pfplsn, pfhpsl, pfhpsn]
    B.check(B.lib.cloudsc2_nl_run(*_host_args(prm, nproma, nlev, ngptot, ptsphy, arrays), C.byref(ms)))
    return ms.value


def cloudsc_driver_tl(prm: B.Params, numomp, nproma, nlev, ngptot, ngptotg, ptsphy, pt, pq, tendency_cml, tendency_loc,
                      pap, paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl, pfplsn, pfhpsl, pfhpsn):
    """CLOUDSC_DRIVER_TL (cloudsc_driver_tl_mod.F90:33-314): returns (znormg[10], passed, itest, kernel_ms)."""
    del numomp, ngptotg
    ms = C.c_double(0.0)
    zn = (C.c_double * 10)()
    arrays = [pt, pq, tendency_cml, tendency_loc, pap, paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl,
              pfplsn, pfhpsl, pfhpsn]
    B.check(B.lib.cloudsc2_tl_taylor_run(*_host_args(prm, nproma, nlev, ngptot, ptsphy, arrays), zn, C.byref(ms)))
    znormg = np.array(zn[:], dtype=np.float64)
    ok, itest = B.taylor_verdict(znormg)
    return znormg, ok, itest, ms.value


def cloudsc_driver_ad(prm: B.Params, numomp, nproma, nlev, ngptot, ngptotg, ptsphy, pt, pq, tendency_cml, tendency_loc,
                      pap, paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl, pfplsn, pfhpsl, pfhpsn):
    """CLOUDSC_DRIVER_AD (cloudsc_driver_ad_mod.F90:22-297): returns (znormg, ok, kernel_ms)."""
    del numomp, ngptotg
    ms = C.c_double(0.0)
    zn = C.c_double(0.0)
    arrays = [pt, pq, tendency_cml, tendency_loc, pap, paph, plu, plude, pmfu, pmfd, pa, pclv, psupsat, pcovptot, pfplsl,
              pfplsn, pfhpsl, pfhpsn]
    B.check(B.lib.cloudsc2_ad_symmetry_run(*_host_args(prm, nproma, nlev, ngptot, ptsphy, arrays), C.byref(zn), C.byref(ms)))
    return zn.value, B.adjoint_verdict(zn.value), ms.value


def run_state(prm: B.Params, st: Cloudsc2State, which: str = "nl"):
    """Convenience: run one of the three drivers on a Cloudsc2State (outputs written into it)."""
    fn = {"nl": cloudsc_driver, "tl": cloudsc_driver_tl, "ad": cloudsc_driver_ad}[which]
    return fn(prm, 1, st.nproma, st.nlev, st.ngptot, st.ngptot, st.ptsphy, *st.driver_arrays())


# ---------------------------------------------------------------------------------------------------------------------
# HBM-resident state
# ---------------------------------------------------------------------------------------------------------------------
def _fld(t, offset_elems: int, stride: int) -> B.Field:
    f = B.Field()
    f.ptr = t.data_ptr() + B.REAL_BYTES * offset_elems
    f.block_stride = stride
    return f


class FlatFields:
    """16 input-shaped or 10 output-shaped device arrays, e.g. the TL increments and the TL outputs: the thread-private scratch
    arrays of the reference's test drivers (cloudsc_driver_tl_mod.F90:77-95) for all blocks at once.

    FlatFields(kind, ...): separate (NBLOCKS, NLEVx, NPROMA) arrays, block stride NLEVx*NPROMA like the state's own arrays.
    FlatFields.pair(...): the 16 inputs AND the 10 outputs of one perturbation / adjoint set; for NPROMA < 64 in ONE buffer
    (NBLOCKS, 26, NLEV+1, NPROMA), field f of block b at plane f -- the AoSoA arrangement the reference itself uses for the
    tendencies.  All 26 fields of a block then lie close together, and a wave's 26 streams share a few entries of the CU's
    address-translation cache instead of occupying 26 (x 2 blocks per wave at NPROMA 32): with separate arrays the TL and AD
    sweeps, which touch 53 and 64 streams per wave at one wave per SIMD, thrash it at small NPROMA (rocprofv3 TCP_UTCL1_*
    counters, profiles/r02_utcl1_counters.txt).  The kernels see fields as pointer + block stride (one stride per layout group, the
    perturbation inputs and outputs of a launch sharing theirs), so both arrangements go through the same launches."""

    NPLANES = len(B.IN_NAMES) + len(B.OUT_NAMES)

    def __init__(self, kind: str, nb: int, nlev: int, nproma: int, device, zero: bool = True, _views=None, _arena=None):
        self.kind = kind
        names = B.IN_NAMES if kind == "in" else B.OUT_NAMES
        half = {"paph"} if kind == "in" else {"fplsl", "fplsn", "fhpsl", "fhpsn"}
        if _views is not None:
            buf, first = _views
            self.buf = buf
            self.t = {n: buf[:, first + k, : nlev + (1 if n in half else 0), :] for k, n in enumerate(names)}
        else:
            shapes = {n: (nb, nlev + (1 if n in half else 0), nproma) for n in names}
            arena = _arena or B.DeviceArena(B.DeviceArena.size_of(shapes.values(), B.REAL_BYTES), device)  # one placed allocation
            self.t = {n: arena.take(shp, zero=zero) for n, shp in shapes.items()}
        self.nlev, self.nproma = nlev, nproma

    @classmethod
    def pair_bytes(cls, nb: int, nlev: int, nproma: int) -> int:
        """Bytes FlatFields.pair takes from an arena (either layout fits)."""
        return B.DeviceArena.size_of([(nb, cls.NPLANES, nlev + 1, nproma)], B.REAL_BYTES) + 32 * B.DeviceArena.ALIGN

    @classmethod
    def pair(cls, nb: int, nlev: int, nproma: int, device, zero: bool = True, arena=None):
        """(inputs, outputs) of one perturbation / adjoint set, interleaved per block in one buffer (see the class docstring);
        CLOUDSC2_SCRATCH_LAYOUT=flat gives two sets of separate arrays instead (A/B measurements)."""
        import os

        # NPROMA < 64: a wave spans two or more blocks (twice the streams): interleaved (TL 2.86 -> 1.66 ms, AD 4.07 -> 3.23 ms at
        # NPROMA 32).  NPROMA >= 64: separate arrays are as fast (TL) or 4 % faster (AD).  The library's own test drivers follow
        # the same rule (csrc/cloudsc2_driver.inc: pair_take).
        layout = os.environ.get("CLOUDSC2_SCRATCH_LAYOUT") or ("blocked" if nproma < 64 else "flat")
        if arena is None:  # `arena`: carve the set out of an existing allocation (DeviceState.from_table(..., reserve=...))
            arena = B.DeviceArena(cls.pair_bytes(nb, nlev, nproma), device)
        if layout == "flat":
            return cls("in", nb, nlev, nproma, device, zero, _arena=arena), cls("out", nb, nlev, nproma, device, zero, _arena=arena)
        shape = (nb, cls.NPLANES, nlev + 1, nproma)
        buf = arena.take(shape, zero=zero)
        return (cls("in", nb, nlev, nproma, device, _views=(buf, 0)),
                cls("out", nb, nlev, nproma, device, _views=(buf, len(B.IN_NAMES))))

    def block(self):
        blk = B.Inputs() if self.kind == "in" else B.Outputs()
        for n, t in self.t.items():
            setattr(blk, n, _fld(t, 0, t.stride(0)))
        return blk

    def zero_(self):
        for t in self.t.values():
            t.zero_()


class DeviceState:
    """GLOBAL_STATE resident in HBM.  torch is used for allocation, streams and H2D/D2H only."""

    FULL = ("PT", "PQ", "PAP", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT", "PCOVPTOT")
    HALF = ("PAPH", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")
    # what the sweeps write (cloudsc2.F90:135-149); inside the arena the read-only arrays come first, like in the library's own state
    WRITTEN = ("PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "B_LOC")

    # the order of the arrays inside the arena = the library's own (csrc/cloudsc2_driver.inc: state_take), which is what
    # cloudsc2_device_malloc_state lays out in every candidate when it times the NL sweep on it
    ORDER = ("PT", "PQ", "PAP", "PLU", "PLUDE", "PMFU", "PMFD", "PSUPSAT", "PAPH", "B_CML", "PCLV",
             "PA", "PCOVPTOT", "QSAT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "B_LOC")

    def _make_arenas(self, read_shapes, written_shapes, reserve: int = 0):
        """The whole state in ONE placed allocation (cloudsc2_device_malloc_state), like the library's own (csrc/cloudsc2_driver.inc:
        state_take): measured on one box, fresh processes, NL at 160 000 columns: 0.812 ms on 8 of 8, against 0.88-0.93 ms with
        the written arrays placed and the read-only ones in a separate (plain or placed) allocation, and 0.90-0.96 ms without
        placement (profiles/r02_placement/z_one_arena_vs_split.txt).  The allocator judges its candidates by the NL sweep on a state
        in exactly this layout (by two generic streams when `reserve` puts perturbation sets behind the state)."""
        self.arena = B.DeviceArena(B.DeviceArena.size_of(list(read_shapes) + list(written_shapes), B.REAL_BYTES) + int(reserve), self.device,
                                   state_geom=(self.nproma, self.nlev, self.ngptot))
        self.arena_in = self.arena

    def __init__(self, st: Cloudsc2State, device="cuda:0"):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.nproma, self.nlev, self.ngptot, self.ptsphy = st.nproma, st.nlev, st.ngptot, st.ptsphy
        self.nb = nblocks_of(st.ngptot, st.nproma)
        shape_of = {n: (st.PT.shape if n == "QSAT" else getattr(st, n).shape) for n in self.ORDER}
        self._make_arenas([shape_of[n] for n in self.ORDER if n not in self.WRITTEN and n != "QSAT"],
                          [shape_of[n] for n in self.ORDER if n in self.WRITTEN or n == "QSAT"])
        for n in self.ORDER:
            d = self.arena.take(shape_of[n], zero=(n == "QSAT"))
            if n != "QSAT":
                d.copy_(torch.from_numpy(getattr(st, n)))
            setattr(self, n, d)
        self._keep = []

    @classmethod
    def from_table(cls, tab: dict, nproma: int, ngptot: int, device="cuda:0", start: int = 0, period: int | None = None,
                   stream=None, reserve: int = 0) -> "DeviceState":
        """CLOUDSC2_ARRAY_STATE_LOAD (cloudsc2_array_state_mod.F90:153-203) without a host copy of the state: the
        KLON-column table is uploaded (a few MB) and tiled into the NPROMA-blocked arrays by cloudsc2_expand_launch;
        outputs are zero-initialised (FIELD_INIT, :186-190).  ``start``/``period``: binding.expand_offsets."""
        import torch

        self = cls.__new__(cls)
        self.torch = torch
        self.device = torch.device(device)
        nlev, klon = tab["PT"].shape
        self.nproma, self.nlev, self.ngptot, self.ptsphy = nproma, nlev, ngptot, float(tab["PTSPHY"])
        self.nb = nblocks_of(ngptot, nproma)
        self._keep = []
        period = klon if period is None else period
        full, half = (self.nb, nlev, nproma), (self.nb, nlev + 1, nproma)
        shape_of = {n: full for n in self.FULL}
        shape_of.update({n: half for n in self.HALF})
        shape_of.update({"B_CML": (self.nb, 8, nlev, nproma), "B_LOC": (self.nb, 8, nlev, nproma), "PCLV": (self.nb, 5, nlev, nproma),
                         "QSAT": full})
        # `reserve` more bytes in the same allocation, e.g. FlatFields.pair_bytes(...) for a perturbation set that is to share the
        # state's placement (FlatFields.pair(..., arena=ds.arena)), the way the library's own test drivers lay their scratch out
        self._make_arenas([shape_of[n] for n in self.ORDER if n not in self.WRITTEN and n != "QSAT"],
                          [shape_of[n] for n in self.ORDER if n in self.WRITTEN or n == "QSAT"], reserve)
        trace = getattr(cls, "_trace", None)  # tools/one_process_series.py: a callable(state, stage) for timing between the stages
        for n in self.ORDER:
            setattr(self, n, self.arena.take(shape_of[n]))
        if trace:
            trace(self, "arrays carved out of the arena, nothing written yet")
        for n in self.ORDER:
            getattr(self, n).zero_()
        if trace:
            trace(self, "all arrays zeroed")
        S, H = nproma * nlev, nproma * (nlev + 1)
        jobs = [(n, getattr(self, n), 0, S if tab[n].shape[0] == nlev else H)
                for n in ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT")]
        jobs += [("TENDENCY_CML_T", self.B_CML, PLANE_T * S, 8 * S), ("TENDENCY_CML_Q", self.B_CML, PLANE_Q * S, 8 * S),
                 ("TENDENCY_CML_QL", self.B_CML, PLANE_QL * S, 8 * S), ("TENDENCY_CML_QI", self.B_CML, PLANE_QI * S, 8 * S),
                 ("PCLV_QL", self.PCLV, 0, 5 * S), ("PCLV_QI", self.PCLV, S, 5 * S)]
        for name, dst, off, stride in jobs:
            src = torch.from_numpy(np.ascontiguousarray(tab[name], dtype=B.REAL)).to(self.device)
            self._keep.append(src)
            B.check(B.lib.cloudsc2_expand_launch(C.cast(src.data_ptr(), _rp), klon, period, start,
                                                 src.shape[0], 1, nproma, ngptot, _fld(dst, off, stride), self._stream(stream)))
        if trace:
            trace(self, "table tiled into the arrays")
        return self

    def validate(self, ref: dict, ngptotg: int | None = None, start: int = 0, period: int | None = None, stream=None):
        """CLOUDSC2_ARRAY_STATE_VALIDATE (cloudsc2_array_state_mod.F90:205-258) on the device, against the KLON-column
        reference table ``ref`` (dataset names of reference.h5), never expanded.  Returns [(name, ndim, stats[5])] in
        the reference's print order and the report text (header + one ERROR_PRINT line per variable)."""
        torch = self.torch
        S, H = self.nproma * self.nlev, self.nproma * (self.nlev + 1)
        ws = torch.empty(B.lib.cloudsc2_validate_workspace_doubles(), dtype=torch.float64, device=self.device)
        dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))  # noqa: E731
        plan = [("PLUDE", self.PLUDE, 0, S, 1), ("PCOVPTOT", self.PCOVPTOT, 0, S, 1), ("PFPLSL", self.PFPLSL, 0, H, 1),
                ("PFPLSN", self.PFPLSN, 0, H, 1), ("PFHPSL", self.PFHPSL, 0, H, 1), ("PFHPSN", self.PFHPSN, 0, H, 1),
                ("TENDENCY_LOC_A", self.B_LOC, PLANE_A * S, 8 * S, 1), ("TENDENCY_LOC_Q", self.B_LOC, PLANE_Q * S, 8 * S, 1),
                ("TENDENCY_LOC_T", self.B_LOC, PLANE_T * S, 8 * S, 1), ("TENDENCY_LOC_CLD", self.B_LOC, PLANE_QL * S, 8 * S, 5)]
        labels = {"TENDENCY_LOC_A": "TENDENCY_LOC%A", "TENDENCY_LOC_Q": "TENDENCY_LOC%Q", "TENDENCY_LOC_T": "TENDENCY_LOC%T",
                  "TENDENCY_LOC_CLD": "TENDENCY_LOC%CLD"}
        rows, keep = [], []
        for name, fld, off, stride, ndim in plan:
            tabdev = torch.from_numpy(np.ascontiguousarray(ref[name], dtype=B.REAL)).to(self.device)
            keep.append(tabdev)
            klon = tabdev.shape[-1]
            nlevx = tabdev.shape[-2]
            stats = torch.empty(5, dtype=torch.float64, device=self.device)
            B.check(B.lib.cloudsc2_validate_launch(C.cast(tabdev.data_ptr(), _rp), klon, klon if period is None else period, start, nlevx, ndim,
                                                   self.nproma, self.ngptot, _fld(fld, off, stride), dp(ws), dp(stats),
                                                   self._stream(stream)))
            rows.append((labels.get(name, name), 2 if ndim == 1 else 3, stats))
        torch.cuda.synchronize(self.device)
        rows = [(n, d, s.cpu().numpy()) for n, d, s in rows]
        nglob = self.ngptot if ngptotg is None else ngptotg
        text = "\n".join([B.validate_header()] + [B.validate_line(n, d, s, nglob) for n, d, s in rows])
        return rows, text

    # -- argument blocks in the driver-array -> kernel-dummy mapping of cloudsc_driver_mod.F90:94-107 --
    def traj_inputs(self, with_qsat: bool = False) -> B.Inputs:
        S, H, P = self.nproma * self.nlev, self.nproma * (self.nlev + 1), self.nproma * self.nlev
        i = B.Inputs()
        i.paph = _fld(self.PAPH, 0, H); i.pap = _fld(self.PAP, 0, S); i.q = _fld(self.PQ, 0, S)
        i.qsat = _fld(self.QSAT, 0, S) if with_qsat else B.Field()
        i.t = _fld(self.PT, 0, S)
        i.l = _fld(self.PCLV, 0 * P, 5 * S); i.i = _fld(self.PCLV, 1 * P, 5 * S)
        i.lude = _fld(self.PLUDE, 0, S); i.lu = _fld(self.PLU, 0, S)
        i.mfu = _fld(self.PMFU, 0, S); i.mfd = _fld(self.PMFD, 0, S)
        i.gtent = _fld(self.B_CML, PLANE_T * P, 8 * S); i.gtenq = _fld(self.B_CML, PLANE_Q * P, 8 * S)
        i.gtenl = _fld(self.B_CML, PLANE_QL * P, 8 * S); i.gteni = _fld(self.B_CML, PLANE_QI * P, 8 * S)
        i.supsat = _fld(self.PSUPSAT, 0, S)
        return i

    def traj_outputs(self) -> B.Outputs:
        S, H, P = self.nproma * self.nlev, self.nproma * (self.nlev + 1), self.nproma * self.nlev
        o = B.Outputs()
        o.tent = _fld(self.B_LOC, PLANE_T * P, 8 * S); o.tenq = _fld(self.B_LOC, PLANE_Q * P, 8 * S)
        o.tenl = _fld(self.B_LOC, PLANE_QL * P, 8 * S); o.teni = _fld(self.B_LOC, PLANE_QI * P, 8 * S)
        o.clc = _fld(self.PA, 0, S); o.covptot = _fld(self.PCOVPTOT, 0, S)
        o.fplsl = _fld(self.PFPLSL, 0, H); o.fplsn = _fld(self.PFPLSN, 0, H)
        o.fhpsl = _fld(self.PFHPSL, 0, H); o.fhpsn = _fld(self.PFHPSN, 0, H)
        return o

    def zero_plane(self) -> B.Field:
        # TENDENCY_LOC(IBL)%cld(:,:,NCLV)=0 (cloudsc_driver_mod.F90:88): plane QV of B_LOC
        S = self.nproma * self.nlev
        return _fld(self.B_LOC, PLANE_QV * S, 8 * S)

    def _stream(self, stream):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    # -- kernel launches (asynchronous on `stream`) --
    def satur(self, prm: B.Params, stream=None):
        i = self.traj_inputs(True)
        B.check(B.lib.cloudsc2_satur_launch(C.byref(prm), self.nproma, self.nlev, self.ngptot, i.pap, i.t, i.qsat,
                                            self._stream(stream)))

    def nl(self, prm: B.Params, stream=None, fused_satur: bool = True, pert_lambda: float = 0.0, outputs: B.Outputs | None = None):
        """SATUR + CLOUDSC2 over all blocks = the body of cloudsc_driver_mod.F90:82-111."""
        i = self.traj_inputs(not fused_satur)
        o = outputs if outputs is not None else self.traj_outputs()
        zp = self.zero_plane() if outputs is None else B.Field()
        B.check(B.lib.cloudsc2_nl_launch(C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, C.byref(i),
                                         C.byref(o), zp, float(pert_lambda), self._stream(stream)))

    def tl(self, prm: B.Params, pert_in: FlatFields | None, pert_out: FlatFields, stream=None, fused_satur: bool = False,
           store_traj: bool = True, supsat_increment: float = 0.01, yy=None):
        """CLOUDSC2TL.  pert_in=None: the increments of the reference's test drivers, dx = 0.01*x (supsat_increment*PSUPSAT for
        PSUPSAT), formed inside the sweep (cloudsc2_tl_launch_self); yy: a float64 tensor of NBLOCKS*NPROMA elements receiving <y,y>
        of every active column then (the adjoint test's norm1)."""
        i = self.traj_inputs(not fused_satur)
        o = self.traj_outputs() if store_traj else B.Outputs()
        do = pert_out.block()
        if pert_in is None:
            B.check(B.lib.cloudsc2_tl_launch_self(C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, C.byref(i),
                                                  C.byref(o), float(supsat_increment), C.byref(do),
                                                  C.c_void_p(yy.data_ptr() if yy is not None else None), self._stream(stream)))
            return
        di = pert_in.block()
        B.check(B.lib.cloudsc2_tl_launch(C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, C.byref(i),
                                         C.byref(o), C.byref(di), C.byref(do), self._stream(stream)))

    def ad(self, prm: B.Params, adj_in: FlatFields, adj_out: FlatFields, scratch, stream=None, fused_satur: bool = False,
           assign: bool = False, sweep: str = "both"):
        """CLOUDSC2AD; assign=True: adj_in = A^T adj_out instead of adj_in += A^T adj_out (cloudsc2_ad_launch_assign).
        sweep = "forward" / "reverse": one of its two sweeps alone (cloudsc2_ad_launch_forward / _reverse; the reverse sweep takes
        PFPLSL5 / PFPLSN5 from the state's flux arrays, where any earlier NL / TL / forward sweep left them).  `scratch` (the cover
        checkpoints) may be None unless LEVAPLS2 / LDRAIN1D is on."""
        i = self.traj_inputs(not fused_satur)
        o = self.traj_outputs()
        sc = C.c_void_p(scratch.data_ptr() if scratch is not None else None)
        geom = (C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, C.byref(i), C.byref(o))
        if sweep == "forward":
            B.check(B.lib.cloudsc2_ad_launch_forward(*geom, sc, self._stream(stream)))
            return
        ai, ao = adj_in.block(), adj_out.block()
        if sweep == "reverse":
            B.check(B.lib.cloudsc2_ad_launch_reverse(*geom, C.byref(ai), C.byref(ao), sc, int(assign), self._stream(stream)))
            return
        if sweep != "both":
            raise ValueError(sweep)
        fn = B.lib.cloudsc2_ad_launch_assign if assign else B.lib.cloudsc2_ad_launch
        B.check(fn(*geom, C.byref(ai), C.byref(ao), sc, self._stream(stream)))

    def adjoint_norms(self, y: FlatFields | None, x_adj: FlatFields | None, norms, blockmax, stream=None):
        """cloudsc2_adjoint_norms_launch: norm1 = <y,y> (y given) and / or norm2, norm3 (x_adj given) per column into norms(3, columns)."""
        i = self.traj_inputs(True)
        q = i.qsat
        yb = y.block() if y is not None else None
        xb = x_adj.block() if x_adj is not None else None
        B.check(B.lib.cloudsc2_adjoint_norms_launch(self.nproma, self.nlev, self.ngptot, C.byref(i) if xb is not None else None,
                                                    C.byref(q) if xb is not None else None, C.byref(yb) if yb is not None else None,
                                                    C.byref(xb) if xb is not None else None, C.c_void_p(norms.data_ptr()),
                                                    C.c_void_p(blockmax.data_ptr()), self._stream(stream)))

    def ad_reverse_norms(self, prm: B.Params, adj_in: FlatFields, adj_out: FlatFields, norms, blockmax, stream=None, fused_satur: bool = False):
        """The adjoint test's AD leg with norm2 / norm3 formed in the reverse sweep (cloudsc2_ad_launch_reverse_norms)."""
        i, o = self.traj_inputs(not fused_satur), self.traj_outputs()
        ai, ao = adj_in.block(), adj_out.block()
        B.check(B.lib.cloudsc2_ad_launch_reverse_norms(C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, C.byref(i), C.byref(o),
                                                       C.byref(ai), C.byref(ao), C.c_void_p(norms.data_ptr()), C.c_void_p(blockmax.data_ptr()),
                                                       self._stream(stream)))

    def taylor_sums(self, pert_outputs: B.Outputs, tl_out: FlatFields, lam: float, stream=None):
        """ERROR_NORM sums of ONE lambda from perturbed outputs in memory (cloudsc2_taylor_sums_launch): (NBLOCKS, 10, 2) doubles."""
        sums = self.torch.zeros((self.nb, 10, 2), dtype=self.torch.float64, device=self.device)
        o, t = self.traj_outputs(), tl_out.block()
        B.check(B.lib.cloudsc2_taylor_sums_launch(self.nproma, self.nlev, self.ngptot, C.byref(o), C.byref(pert_outputs), C.byref(t),
                                                  float(lam), C.c_void_p(sums.data_ptr()), self._stream(stream)))
        return sums

    def taylor_sweep(self, prm: B.Params, tl_out: FlatFields, nproma_stat: int | None = None, fused_satur: bool = False, stream=None):
        """The whole lambda loop of the Taylor test in one sweep, the lambdas on the lanes (cloudsc2_taylor_sweep_launch):
        (10 lambdas, blocks of the statistic, 10 fields, 2) doubles.  The state's outputs must hold the base run."""
        nps = int(nproma_stat or self.nproma)
        nbs = (self.ngptot + nps - 1) // nps
        n = C.c_longlong(0)
        B.check(B.lib.cloudsc2_taylor_sweep_work_doubles(self.nproma, self.ngptot, C.byref(n)))
        # The sweep's workspace belongs to this state, one per stream it was ever launched on, and lives as long as the state: the
        # launch is asynchronous on a stream torch's caching allocator may know nothing about (a raw hipStream_t), so a tensor
        # dropped here could be handed out again while the sweep or the reduce kernel still reads it.
        sh = self._stream(stream)
        key = (int(sh.value or 0) if hasattr(sh, "value") else int(sh or 0), int(n.value))
        cache = self.__dict__.setdefault("_taylor_work", {})
        work = cache.get(key)
        if work is None:
            work = cache[key] = self.torch.empty((n.value,), dtype=self.torch.float64, device=self.device)
        sums = self.torch.zeros((10, nbs, 10, 2), dtype=self.torch.float64, device=self.device)
        i, o, t = self.traj_inputs(not fused_satur), self.traj_outputs(), tl_out.block()
        B.check(B.lib.cloudsc2_taylor_sweep_launch(C.byref(prm), self.ptsphy, self.nproma, self.nlev, self.ngptot, nps, C.byref(i), C.byref(o),
                                                   C.byref(t), C.c_void_p(work.data_ptr()), C.c_void_p(sums.data_ptr()), sh))
        return sums

    def increments(self, zero_supsat: bool = False, into: FlatFields | None = None) -> FlatFields:
        """dx = 0.01 * x for the 16 inputs (cloudsc_driver_tl_mod.F90:156-171); ZSUPSAT = 0 in the adjoint test
        (cloudsc_driver_ad_mod.F90:139).  `into`: an existing input set (e.g. one half of FlatFields.pair) to fill."""
        ff = into if into is not None else FlatFields("in", self.nb, self.nlev, self.nproma, self.device, zero=False)
        src = {"paph": self.PAPH, "pap": self.PAP, "q": self.PQ, "qsat": self.QSAT, "t": self.PT,
               "l": self.PCLV[:, 0], "i": self.PCLV[:, 1], "lude": self.PLUDE, "lu": self.PLU, "mfu": self.PMFU,
               "mfd": self.PMFD, "gtent": self.B_CML[:, PLANE_T], "gtenq": self.B_CML[:, PLANE_Q],
               "gtenl": self.B_CML[:, PLANE_QL], "gteni": self.B_CML[:, PLANE_QI], "supsat": self.PSUPSAT}
        for n, s in src.items():
            if n == "supsat" and zero_supsat:
                ff.t[n].zero_()
            else:
                self.torch.mul(s, 0.01, out=ff.t[n])
        return ff

    def new_scratch(self):
        return B.device_empty((self.nb, self.nlev, self.nproma), device=self.device)

    def download(self, st: Cloudsc2State) -> Cloudsc2State:
        for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN"):
            getattr(st, n)[...] = getattr(self, n).cpu().numpy()
        return st


class ResidentState:
    """The library-owned resident GLOBAL_STATE (cloudsc2_state_* of include/cloudsc2_hip.h): what the Fortran mains use with
    CLOUDSC2_RESIDENT=1.  No torch: tables go in as numpy arrays, statistics and verdict norms come back as numbers."""

    TABLE_FIELDS = {"PT": 0, "PQ": 1, "PAP": 2, "PAPH": 3, "PLU": 4, "PLUDE": 5, "PMFU": 6, "PMFD": 7, "PA": 8, "PSUPSAT": 9,
                    "TENDENCY_CML_T": B.F_CML_T + PLANE_T, "TENDENCY_CML_Q": B.F_CML_T + PLANE_Q,
                    "TENDENCY_CML_QL": B.F_CML_T + PLANE_QL, "TENDENCY_CML_QI": B.F_CML_T + PLANE_QI,
                    "PCLV_QL": B.F_PCLV_QL, "PCLV_QI": B.F_PCLV_QL + 1}

    def __init__(self, nproma: int, nlev: int, ngptot: int):
        self.h = C.c_void_p()
        B.check(B.lib.cloudsc2_state_create(int(nproma), int(nlev), int(ngptot), C.byref(self.h)))
        self.nproma, self.nlev, self.ngptot = nproma, nlev, ngptot

    def __del__(self):
        lib = getattr(B, "lib", None) if B is not None else None  # (module globals are gone at interpreter shutdown)
        if lib is not None and getattr(self, "h", None) is not None and self.h.value:
            lib.cloudsc2_state_destroy(self.h)
            self.h.value = None

    @classmethod
    def from_table(cls, tab: dict, nproma: int, ngptot: int, start: int = 0, period: int | None = None) -> "ResidentState":
        nlev, klon = tab["PT"].shape
        self = cls(nproma, nlev, ngptot)
        self.ptsphy = float(tab["PTSPHY"])
        for name, fid in cls.TABLE_FIELDS.items():
            t = np.ascontiguousarray(tab[name], dtype=B.REAL)
            B.check(B.lib.cloudsc2_state_expand(self.h, fid, _ptr(t), klon, klon if period is None else period, start))
        return self

    def blocking(self):
        """(NPROMA of the device arrays -- the library's choice --, the caller's NPROMA)"""
        d, u = C.c_int(), C.c_int()
        B.check(B.lib.cloudsc2_state_blocking(self.h, C.byref(d), C.byref(u)))
        return d.value, u.value

    def nl(self, prm: B.Params, repeats: int = 1) -> float:
        ms = C.c_double()
        B.check(B.lib.cloudsc2_state_nl(self.h, C.byref(prm), self.ptsphy, int(repeats), C.byref(ms)))
        return ms.value

    def tl_taylor(self, prm: B.Params):
        zn, ms = (C.c_double * 10)(), C.c_double()
        B.check(B.lib.cloudsc2_state_tl_taylor(self.h, C.byref(prm), self.ptsphy, zn, C.byref(ms)))
        z = np.array(zn[:])
        ok, itest = B.taylor_verdict(z)
        return z, ok, itest, ms.value

    def ad_symmetry(self, prm: B.Params):
        zn, ms = C.c_double(), C.c_double()
        B.check(B.lib.cloudsc2_state_ad_symmetry(self.h, C.byref(prm), self.ptsphy, C.byref(zn), C.byref(ms)))
        return zn.value, B.adjoint_verdict(zn.value), ms.value

    def validate(self, field: int, ref_table: np.ndarray, ndim: int = 1, start: int = 0, period: int | None = None) -> np.ndarray:
        t = np.ascontiguousarray(ref_table, dtype=B.REAL)
        klon = t.shape[-1]
        st = (C.c_double * 5)()
        B.check(B.lib.cloudsc2_state_validate(self.h, int(field), int(ndim), _ptr(t), klon, klon if period is None else period, start, st))
        return np.array(st[:])

    def download(self, st: Cloudsc2State) -> Cloudsc2State:
        B.check(B.lib.cloudsc2_state_download(self.h, _ptr(st.B_LOC), _ptr(st.PA), _ptr(st.PCOVPTOT), _ptr(st.PFPLSL), _ptr(st.PFPLSN),
                                              _ptr(st.PFHPSL), _ptr(st.PFHPSN)))
        return st

    def upload(self, st: Cloudsc2State):
        B.check(B.lib.cloudsc2_state_upload(self.h, *[_ptr(a) for a in st.driver_arrays()]))
