"""GLOBAL_STATE of the CLOUDSC2 dwarf in the reference's NPROMA-blocked layout, plus the synthetic atmosphere.

Layout (reference: src/common/module/cloudsc2_array_state_mod.F90:26-151).  Fortran ``F(NPROMA, NLEV, NBLOCKS)``
is held here as a C-ordered numpy/torch array of shape ``(NBLOCKS, NLEV, NPROMA)`` -- the same bytes.  The AoSoA
tendency buffers ``B_CML/B_LOC(NPROMA, NLEV, 8, NBLOCKS)`` are ``(NBLOCKS, 8, NLEV, NPROMA)`` with plane order
T=0, A=1, Q=2, CLD(QL, QI, QR, QS, QV)=3..7 (:145-148); ``PCLV(NPROMA, NLEV, 5, NBLOCKS)`` is
``(NBLOCKS, 5, NLEV, NPROMA)``.

``config-files/input.h5`` is not distributed with the reference checkout (``.MISSING_LARGE_BLOBS``), so the
inputs are a deterministic synthetic atmosphere of 100 distinct columns that is tiled periodically
to NGPTOT columns exactly like ``expand_r2`` (src/common/module/expand_mod.F90:270-302): column g (0-based) holds
table column ``g mod 100``; the tail of the last block is zero.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

KLON_TABLE = 100  # number of distinct columns, = KLON of the reference's input.h5 / reference.h5
NLEV_DEFAULT = 137

# plane indices inside B_CML / B_LOC and PCLV (0-based)
PLANE_T, PLANE_A, PLANE_Q, PLANE_QL, PLANE_QI, PLANE_QR, PLANE_QS, PLANE_QV = range(8)
NCLDQL, NCLDQI = 0, 1


def nblocks_of(ngptot: int, nproma: int) -> int:
    # NGPBLKS = (NGPTOT / NPROMA) + MIN(MOD(NGPTOT,NPROMA), 1)   (cloudsc_driver_mod.F90:64)
    return ngptot // nproma + min(ngptot % nproma, 1)


def column_range(ngptotg: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous sub-range of the global columns owned by ``rank`` -- the reference's MPI split
    (src/cloudsc2_nl/dwarf_cloudsc.F90:66-69): NGPTOT = ceil(NGPTOTG/NUMPROC), last rank takes the remainder."""
    per = (ngptotg - 1) // world + 1
    start = min(rank * per, ngptotg)
    stop = min(start + per, ngptotg)
    return start, stop


def synthetic_table(nlev: int = NLEV_DEFAULT, ncol: int = KLON_TABLE, consts: dict | None = None) -> dict:
    """100 distinct columns, each field shaped (NLEV or NLEV+1, ncol) like the (KLEV, KLON) datasets of input.h5.

    Recipe of SURVEY.md 8d: every column carries cloud and precipitates (the reference's Taylor test STOPs on a
    block without active statistics), none is near-trivial (the adjoint test is relative per column).

    The values come from the library's ``cloudsc2_synthetic_table`` (host code, include/cloudsc2_hip.h) -- the ONE
    implementation the Fortran mains load as well, so that every front end runs the same bits: the
    Taylor test's verdict is decided by round-off, and numpy's, flang's and glibc's exp / pow differ in the last place.
    """
    import ctypes as C

    from . import binding as B

    c = consts or {}
    rd, rv, rtt = c.get("rd", 287.0597), c.get("rv", 461.5250), c.get("rtt", 273.16)
    names = ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PCLV_QL", "PCLV_QI", "TENDENCY_CML_T", "TENDENCY_CML_Q")
    arr = {n: np.zeros((nlev + (1 if n == "PAPH" else 0), ncol)) for n in names}
    dp = C.POINTER(C.c_double)
    B.check(B.lib.cloudsc2_synthetic_table(int(ncol), int(nlev), float(rd), float(rv), float(rtt), *[arr[n].ctypes.data_as(dp) for n in names]))
    zeros = np.zeros((nlev, ncol))
    return {**arr, "PA": zeros.copy(), "PSUPSAT": zeros.copy(), "TENDENCY_CML_QL": zeros.copy(), "TENDENCY_CML_QI": zeros.copy(),
            "PTSPHY": 3600.0}


def random_table(nlev: int, ncol: int, seed: int) -> dict:
    """Randomised variant of the synthetic atmosphere for parity tests: same structure, seeded perturbations of
    every field (including non-zero PSUPSAT and cloud tendencies) so that more branch combinations are visited."""
    rng = np.random.default_rng(seed)
    tab = synthetic_table(nlev, ncol)
    u = lambda shape, a, b: rng.uniform(a, b, size=shape)  # noqa: E731
    tab["PT"] = tab["PT"] + u((nlev, ncol), -3.0, 3.0)
    tab["PQ"] = tab["PQ"] * u((nlev, ncol), 0.7, 1.25)
    tab["PCLV_QL"] = tab["PCLV_QL"] * u((nlev, ncol), 0.2, 3.0)
    tab["PCLV_QI"] = tab["PCLV_QI"] * u((nlev, ncol), 0.2, 3.0)
    tab["PLU"] = tab["PLU"] * u((nlev, ncol), 0.5, 1.5)
    tab["PLUDE"] = tab["PLUDE"] * u((nlev, ncol), 0.0, 2.0)
    tab["PMFU"] = tab["PMFU"] * u((nlev, ncol), 0.5, 1.5)
    tab["PMFD"] = tab["PMFD"] * u((nlev, ncol), 0.5, 1.5)
    tab["PSUPSAT"] = tab["PQ"] * u((nlev, ncol), 0.0, 1e-3)
    tab["TENDENCY_CML_T"] = tab["TENDENCY_CML_T"] * u((nlev, ncol), -1.0, 2.0)
    tab["TENDENCY_CML_Q"] = tab["TENDENCY_CML_Q"] * u((nlev, ncol), -1.0, 2.0)
    tab["TENDENCY_CML_QL"] = tab["PCLV_QL"] * u((nlev, ncol), -1e-5, 1e-5)
    tab["TENDENCY_CML_QI"] = tab["PCLV_QI"] * u((nlev, ncol), -1e-5, 1e-5)
    # surface pressure varies per column; keep PAP between the half levels
    scale = u((1, ncol), 0.93, 1.03)
    tab["PAPH"] = tab["PAPH"] * scale
    tab["PAP"] = tab["PAP"] * scale
    return tab


def ceta_from_table(tab: dict) -> np.ndarray:
    # YRECLD%CETA(JK) = PAP(1,JK,1)/PAPH(1,KLEV+1,1): column 1 of block 1 only (dwarf_cloudsc.F90:100-102)
    return np.ascontiguousarray(tab["PAP"][:, 0] / tab["PAPH"][-1, 0])


def _tile(field2d: np.ndarray, nproma: int, ngptot: int, col0: int = 0) -> np.ndarray:
    """(NLEVx, ncol) table -> (NBLOCKS, NLEVx, NPROMA), periodic in the global column index (expand_mod.F90:283-296);
    ``col0`` is the global index of this rank's first column."""
    nlevx, ncol = field2d.shape
    nb = nblocks_of(ngptot, nproma)
    out = np.zeros((nb * nproma, nlevx), dtype=np.float64)
    idx = (col0 + np.arange(ngptot)) % ncol
    out[:ngptot, :] = field2d.T[idx, :]
    return np.ascontiguousarray(out.reshape(nb, nproma, nlevx).transpose(0, 2, 1))


@dataclasses.dataclass
class Cloudsc2State:
    """Host-side GLOBAL_STATE (numpy; fp64, or fp32 under CLOUDSC2_PRECISION=single).  Field names are the reference's."""

    nproma: int
    nlev: int
    ngptot: int
    ptsphy: float
    PT: np.ndarray
    PQ: np.ndarray
    B_CML: np.ndarray
    B_LOC: np.ndarray
    PAP: np.ndarray
    PAPH: np.ndarray
    PLU: np.ndarray
    PLUDE: np.ndarray
    PMFU: np.ndarray
    PMFD: np.ndarray
    PA: np.ndarray
    PCLV: np.ndarray
    PSUPSAT: np.ndarray
    PCOVPTOT: np.ndarray
    PFPLSL: np.ndarray
    PFPLSN: np.ndarray
    PFHPSL: np.ndarray
    PFHPSN: np.ndarray

    @property
    def nblocks(self) -> int:
        return nblocks_of(self.ngptot, self.nproma)

    DRIVER_ORDER = ("PT", "PQ", "B_CML", "B_LOC", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PCLV",
                    "PSUPSAT", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")

    def driver_arrays(self):
        """Arrays in the argument order of CLOUDSC_DRIVER (cloudsc_driver_mod.F90:22-30)."""
        return [getattr(self, n) for n in self.DRIVER_ORDER]

    def copy(self) -> "Cloudsc2State":
        kw = {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}
        for k, v in kw.items():
            if isinstance(v, np.ndarray):
                kw[k] = v.copy()
        return Cloudsc2State(**kw)

    # output fields the reference validates (cloudsc2_array_state_mod.F90:246-256), as flat views
    def outputs(self) -> dict:
        return {
            "PCOVPTOT": self.PCOVPTOT, "PFPLSL": self.PFPLSL, "PFPLSN": self.PFPLSN, "PFHPSL": self.PFHPSL,
            "PFHPSN": self.PFHPSN, "PA": self.PA,
            "TENDENCY_LOC_T": self.B_LOC[:, PLANE_T], "TENDENCY_LOC_Q": self.B_LOC[:, PLANE_Q],
            "TENDENCY_LOC_QL": self.B_LOC[:, PLANE_QL], "TENDENCY_LOC_QI": self.B_LOC[:, PLANE_QI],
        }


def state_from_table(tab: dict, nproma: int, ngptot: int, col0: int = 0, poison_outputs: float | None = None,
                     real=None) -> Cloudsc2State:
    """LOAD of cloudsc2_array_state_mod.F90:153-203 with the table standing in for input.h5: tile the inputs
    (LOAD_AND_EXPAND, :167-182) and zero-initialise the outputs (FIELD_INIT, :186-190)."""
    nlev = tab["PT"].shape[0]
    nb = nblocks_of(ngptot, nproma)
    t = lambda name: _tile(tab[name], nproma, ngptot, col0)  # noqa: E731
    b_cml = np.zeros((nb, 8, nlev, nproma))
    b_cml[:, PLANE_T] = t("TENDENCY_CML_T")
    b_cml[:, PLANE_Q] = t("TENDENCY_CML_Q")
    b_cml[:, PLANE_QL] = t("TENDENCY_CML_QL")
    b_cml[:, PLANE_QI] = t("TENDENCY_CML_QI")
    pclv = np.zeros((nb, 5, nlev, nproma))
    pclv[:, NCLDQL] = t("PCLV_QL")
    pclv[:, NCLDQI] = t("PCLV_QI")
    fill = 0.0 if poison_outputs is None else poison_outputs
    full = lambda: np.full((nb, nlev, nproma), fill)  # noqa: E731
    half = lambda: np.full((nb, nlev + 1, nproma), fill)  # noqa: E731
    if real is None:
        from .binding import REAL as real  # the precision of the loaded library (CLOUDSC2_PRECISION)
    if np.dtype(real) != np.float64:  # JPRB = fp32: the fp64 file data are rounded on load, as under the reference's -DSINGLE
        t0, full0, half0 = t, full, half
        t = lambda name: t0(name).astype(real)  # noqa: E731
        full = lambda: full0().astype(real)  # noqa: E731
        half = lambda: half0().astype(real)  # noqa: E731
        b_cml, pclv = b_cml.astype(real), pclv.astype(real)
    return Cloudsc2State(
        nproma=nproma, nlev=nlev, ngptot=ngptot, ptsphy=float(tab["PTSPHY"]),
        PT=t("PT"), PQ=t("PQ"), B_CML=b_cml, B_LOC=np.full((nb, 8, nlev, nproma), fill, dtype=real),
        PAP=t("PAP"), PAPH=t("PAPH"), PLU=t("PLU"), PLUDE=t("PLUDE"), PMFU=t("PMFU"), PMFD=t("PMFD"),
        PA=t("PA") if poison_outputs is None else full(), PCLV=pclv, PSUPSAT=t("PSUPSAT"),
        PCOVPTOT=full(), PFPLSL=half(), PFPLSN=half(), PFHPSL=half(), PFHPSN=half(),
    )


def bytes_per_column(nlev: int, kernel: str = "nl", real_bytes: int | None = None) -> int:
    """Algorithmic HBM bytes per column: every input plane read once, every output plane written once."""
    if real_bytes is None:
        from .binding import REAL_BYTES as real_bytes
    return (real_bytes * _reals_per_column(nlev, kernel))


def _reals_per_column(nlev: int, kernel: str) -> int:
    full, half = nlev, nlev + 1
    nl_in = half + 14 * full                 # PAPH + 14 full-level planes (PQS comes from the fused SATUR)
    nl_out = 6 * full + 4 * half             # PTEN{T,Q,L,I}, PCLC, PCOVPTOT + 4 flux planes
    if kernel == "nl":
        return nl_in + nl_out
    if kernel == "nl_driver":                # + the driver's CLD(:,:,NCLV)=0 plane
        return nl_in + nl_out + full
    if kernel == "tl":
        return nl_in + full + (half + 15 * full) + 2 * nl_out
    x = half + 15 * full
    if kernel == "ad":                        # traj in (+PQS5), y in, x in (+=), y <- 0, x out, traj out
        return (nl_in + full) + nl_out + x + nl_out + x + nl_out
    if kernel == "ad_design_floor":           # what a two-pass adjoint of a 137-level column must move: `ad` + the SECOND read of the
        # trajectory inputs by the reverse pass (cloudsc2ad.F90:366-866 forward, :877-1740 reverse: the reference keeps the whole
        # trajectory of a block in cache-resident arrays; at 17.5 KB per column it does not survive on chip for a CU's 256 columns)
        return (nl_in + full) + nl_out + x + nl_out + x + nl_out + (nl_in + full)
    if kernel == "ad_reverse":                # reverse sweep alone: traj in (+PQS5), PFPLSL5/PFPLSN5 at JK, y in, x in, y <- 0, x out
        return (nl_in + full) + 2 * full + nl_out + x + nl_out + x
    if kernel == "ad_ckpt":                   # the cover-checkpoint plane, written and re-read (evaporation branch only)
        return 2 * full
    if kernel == "ad_old_adjoints":           # the 16 old input adjoints, which the assign forms do not read
        return x
    raise ValueError(kernel)


def validate_l1(ref: np.ndarray, got: np.ndarray) -> float:
    """The reference's acceptance number per field: sum|err| / sum|ref| (validate_mod.F90:274-289); a field is flagged
    when it exceeds 10*eps."""
    den = float(np.sum(np.abs(ref)))
    num = float(np.sum(np.abs(got - ref)))
    if den == 0.0:
        return 0.0 if num == 0.0 else math.inf
    return num / den
