"""The data formats either side of the hot path pinned by the REFERENCE ITSELF: the tiler against
EXPAND_R2 / EXPAND_R3 and GET_OFFSETS of expand_mod.F90, the validation report against VALIDATE_R2 / R3 and ERROR_PRINT of
validate_mod.F90, the drivers' timing table against PERFORMANCE_TIMER%PRINT_PERFORMANCE of timer_mod.F90 -- all three modules
compiled unmodified into oracle/_ref/libcloudsc2_ref.so by oracle/Makefile and reached through oracle/ref_harness.F90.
(The host restatements these formats were checked against before are now themselves checked against the reference.)"""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

from tests.util import ROOT, c2, refcall

pytestmark = pytest.mark.skipif(not refcall.have_ref(), reason="oracle/_ref/libcloudsc2_ref.so not built")


@pytest.fixture(scope="module")
def ref():
    return refcall.RefLib()


def rank_columns(ngptotg, irank, numproc):
    """the mains' split of NGPTOTG over the ranks (dwarf_cloudsc.F90:64-69)"""
    n = (ngptotg - 1) // numproc + 1
    return ngptotg - (numproc - 1) * n if irank == numproc - 1 else n


# (NLON of the file, NGPTOTG, rank, ranks): the table covers the global domain (rank offsets apply) or is tiled (no offsets)
SPLITS = [(100, 160000, 0, 1), (100, 160000, 3, 8), (100, 100, 2, 4), (100, 100, 3, 4), (100, 50, 1, 4), (100, 100, 1, 2),
          (100, 100, 2, 3), (100, 64, 0, 1), (7, 7, 0, 1), (100, 1280000, 7, 8), (100, 90, 3, 4)]
CASES = [(nlon, rank_columns(g, r, p), g, r, p) for nlon, g, r, p in SPLITS]


@pytest.mark.parametrize("nlon,ngptot,ngptotg,irank,numproc", CASES)
def test_expand_offsets_equal_get_offsets(ref, nlon, ngptot, ngptotg, irank, numproc):
    """cloudsc2_expand_offsets = GET_OFFSETS (expand_mod.F90:30-46): 0-based start, period = size."""
    start, end, size = ref.get_offsets(irank, numproc, nlon, ngptot, ngptotg)
    assert c2.binding.expand_offsets(nlon, ngptot, ngptotg, irank, numproc) == (start - 1, size)
    assert end == start + size - 1


@pytest.mark.parametrize("nproma,ngptot", [(32, 250), (100, 100), (128, 1000), (64, 70), (2, 7), (16, 100), (256, 1000), (100, 350)])
def test_host_tiling_equals_expand_r2_r3(ref, nproma, ngptot):
    """state_from_table (what the GPU tests hold the device tiler to) = EXPAND_R2 / EXPAND_R3 (expand_mod.F90:270-335), bit
    for bit, for even NPROMA (the reference's own restriction: MOD(GIDX,NLON) = 0 is mishandled for odd NPROMA) and ragged
    tails; the padded tail of the last block is zero in both."""
    tab = c2.random_table(137, 100, seed=21)
    st = c2.state_from_table(tab, nproma, ngptot)
    for name in ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT"):
        assert np.array_equal(getattr(st, name), ref.expand(tab[name], nproma, ngptot)), name
    pclv = np.stack([tab["PCLV_QL"], tab["PCLV_QI"], np.zeros_like(tab["PT"]), np.zeros_like(tab["PT"]), np.zeros_like(tab["PT"])])
    assert np.array_equal(st.PCLV, ref.expand(pclv, nproma, ngptot))


@pytest.mark.parametrize("nlon,ngptot,ngptotg,irank,numproc", [c for c in CASES if c[1] <= 20000])
def test_rank_slices_tile_like_load_and_expand(ref, nlon, ngptot, ngptotg, irank, numproc):
    """A rank of a multi-rank run: LOAD_AND_EXPAND_R2 (expand_mod.F90:101-116) reads table columns START..END and expands
    them with NLON = SIZE; the library's (start, period) pair addresses the same columns: field(g) = table((start + g mod
    period) mod NLON) -- what cloudsc2_expand_launch computes on the device (tests/test_gpu_io.py checks the device against this)."""
    rng = np.random.default_rng(5)
    table = rng.standard_normal((5, nlon))
    start, end, size = ref.get_offsets(irank, numproc, nlon, ngptot, ngptotg)
    nproma = 16
    want = ref.expand(table[:, start - 1:end], nproma, ngptot)
    s0, period = c2.binding.expand_offsets(nlon, ngptot, ngptotg, irank, numproc)
    cols = (s0 + np.arange(ngptot) % period) % nlon
    nb = (ngptot + nproma - 1) // nproma
    got = np.zeros((nb * nproma, 5))
    got[:ngptot] = table.T[cols]
    assert np.array_equal(want, got.reshape(nb, nproma, 5).transpose(0, 2, 1))


def test_validate_format_equals_error_print(ref):
    """cloudsc2_validate_format = the line ERROR_PRINT writes (validate_mod.F90:263-296), byte for byte: option codes 1/2/3, the
    10 eps `!!!!` rule, E20.13 editing incl. negative zero, exponent carries and three-digit exponents."""
    rng = np.random.default_rng(11)
    eps = np.finfo(np.float64).eps
    cases = [("PCOVPTOT", 2, [0.0, 1.0, 0.0, 0.0, 5.0e3], 100), ("PFPLSL", 2, [0.0, 3.9e-4, 1e-12, 4e-10, 1e-20], 100),
             ("TENDENCY_LOC%CLD", 3, [-2.5e-7, 3.0e-7, 1e-22, 3e-16, 2.0], 160000),
             ("PFHPSN", 2, [-1.2345678901234567e3, 0.0, 2.5e-9, 1e-6, 1e5], 100), ("X", 2, [-0.0, 0.0, 0.0, 0.0, 0.0], 1),
             ("X", 2, [9.9999999999999995e-8, 1e300, 1e-300, 0.0, 0.0], 1), ("TENDENCY_LOC%T", 2, [-1e-3, 1e-3, 1e-19, 10 * eps * 7.0, 7.0], 100),
             ("TENDENCY_LOC%Q", 2, [-1e-3, 1e-3, 1e-19, 10.000001 * eps * 7.0, 7.0], 100), ("A_NAME_LONGER_THAN_TWENTY_CHARS", 2, [1, 2, 3, 4, 5], 7)]
    for _ in range(200):
        mag = 10.0 ** rng.uniform(-30, 30, 5)
        s = [-mag[0], mag[1], mag[2], mag[3] * rng.choice([1.0, 1e-20]), mag[4] * rng.choice([1.0, 1e-25])]
        cases.append((f"F{_}", int(rng.integers(1, 4)), s, int(rng.integers(1, 2000000))))
    for name, ndim, s, n in cases:
        want = ref.error_print(name, s[0], s[1], s[2], s[3], s[4], s[3] / n, ndim)
        assert c2.binding.validate_line(name, ndim, s, n) == want, (name, s, n)
    assert c2.binding.validate_header() == " " + "Variable".rjust(20) + " Dim" + "".join(
        " " + h.rjust(20) for h in ("MinValue", "MaxValue", "AbsMaxErr", "AvgAbsErr/GP", "MaxRelErr-%"))  # print of cloudsc2_array_state_mod.F90:244


def _print_perf_binary(tmp_path_factory):
    fdir = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran")
    bld = os.path.join(fdir, "build")
    if not os.path.exists(os.path.join(bld, "cloudsc_driver_mod.o")):
        subprocess.check_call(["make", "-C", fdir], stdout=subprocess.DEVNULL)
    out = str(tmp_path_factory.mktemp("print_perf") / "print_perf")
    csrc = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "csrc")
    objs = [os.path.join(bld, f) for f in ("cloudsc2_support.o", "cloudsc_mpi_mod.o", "cloudsc2_hip_mod.o", "cloudsc_driver_mod.o")]
    subprocess.check_call(["/opt/rocm/bin/amdflang", "-cpp", "-O2", "-module-dir", os.path.dirname(out), f"-I{bld}",
                           os.path.join(ROOT, "tests", "fortran", "print_perf.F90"), *objs, f"-L{csrc}", "-lcloudsc2_hip", "-lcloudsc2_io",
                           "-lcloudsc2_comm", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", out],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/amdflang"), reason="no Fortran compiler")
def test_timing_table_equals_the_references(ref, tmp_path_factory):
    """CLOUDSC2_PRINT_PERFORMANCE of the Fortran drivers writes PERFORMANCE_TIMER%PRINT_PERFORMANCE's table (timer_mod.F90:114-174,
    formats 1000-1003, whole milliseconds) byte for byte when fed what one worker thread of the reference would have measured;
    the one line it adds below the table carries the sub-millisecond kernel time."""
    exe = _print_perf_binary(tmp_path_factory)
    for nproma, ngptot, kernel_ms, wall_s, dev in [(32, 160000, 97.1, 0.0975, 3), (128, 160000, 0.82, 0.0613, 0), (100, 100, 0.0, 0.0, 5),
                                                  (128, 1048576, 4.9, 0.4021, 7), (1, 100, 12.7, 0.0139, 1)]:
        nblk = (ngptot + nproma - 1) // nproma
        r = subprocess.run([exe, "1", str(nproma), str(nblk), str(ngptot), repr(kernel_ms), repr(wall_s), str(dev)], capture_output=True,
                           text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        lines = r.stderr.splitlines()
        want = ref.timer_print(nproma, nblk, ngptot, [kernel_ms * 1e-3], [dev], [nblk], [ngptot], wall_s).splitlines()
        assert lines[:4] == want and len(want) == 4, (lines, want)
        assert lines[4].startswith(" GPU kernel") and "columns/s" in lines[4] and len(lines) == 5
