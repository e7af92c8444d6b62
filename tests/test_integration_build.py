"""INTEGRATION.md steps 2-3 kept true by a test: the binding module and the three driver modules of fortran/ compile
against the REFERENCE's own module files (parkind1, yomphyder, yomcst, yoethf, yoecldp, yoephli, yoecld, yophnc, yomncl
as built from /root/reference by oracle/Makefile into oracle/_ref/build) -- i.e. they can replace the reference's
cloudsc_driver*_mod.F90 inside the reference's source tree without the stand-in modules of fortran/support/."""
from __future__ import annotations

import os
import subprocess

import pytest

from tests.util import ROOT

FC = "/opt/rocm/bin/amdflang"
REFMODS = os.path.join(ROOT, "oracle", "_ref", "build")
FDIR = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran")


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not installed")
@pytest.mark.skipif(not os.path.exists(os.path.join(REFMODS, "yomphyder.mod")), reason="oracle/_ref/build not built (needs /root/reference)")
@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_drivers_compile_against_the_reference_modules(tmp_path, prec):
    refmods = REFMODS if prec == "dp" else os.path.join(ROOT, "oracle", "_ref", "build_sp")
    if not os.path.exists(os.path.join(refmods, "yomphyder.mod")):
        pytest.skip(f"{refmods} not built")
    flags = ["-cpp", "-O2", "-fPIC"] + (["-DSINGLE"] if prec == "sp" else [])
    # cloudsc_mpi_mod.F90 (RCCL) replaces the reference's module of the same name; the drivers below then USE it
    for src in ("cloudsc_mpi_mod.F90", "cloudsc2_hip_mod.F90", "cloudsc_driver_mod.F90", "cloudsc_driver_tl_mod.F90", "cloudsc_driver_ad_mod.F90"):
        # our modules go to tmp_path (searched first); everything else they USE must come from the reference's build
        cmd = [FC, *flags, "-module-dir", str(tmp_path), "-I", str(tmp_path), "-I", refmods, "-c", os.path.join(FDIR, src),
               "-o", str(tmp_path / (src[:-4] + ".o"))]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (" ".join(cmd), r.stderr[-3000:])
    # the object files define the reference's entry points (module procedure symbols of flang)
    nm = subprocess.run(["nm", str(tmp_path / "cloudsc_driver_mod.o")], capture_output=True, text=True).stdout.lower()
    assert "cloudsc_driver" in nm and "cloudsc2_nl_run" in nm
    # none of the stand-in modules was used: they are not on the include path, and no .mod of them was produced
    produced = {f for f in os.listdir(tmp_path) if f.endswith(".mod")}
    assert produced == {"cloudsc_mpi_mod.mod", "cloudsc2_hip_mod.mod", "cloudsc_driver_mod.mod", "cloudsc_driver_tl_mod.mod",
                        "cloudsc_driver_ad_mod.mod"}, produced


@pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not installed")
@pytest.mark.skipif(not os.path.exists(os.path.join(REFMODS, "cloudsc_mpi_mod.mod")), reason="oracle/_ref/build not built")
def test_drivers_also_compile_against_the_references_own_mpi_module(tmp_path):
    """Without fortran/cloudsc_mpi_mod.F90 on the path the drivers pick up the REFERENCE's cloudsc_mpi_mod (the no-MPI build:
    NUMPROC = 1, dummy reductions): a maintainer can adopt the HIP drivers first and the RCCL module later."""
    for src in ("cloudsc2_hip_mod.F90", "cloudsc_driver_mod.F90", "cloudsc_driver_tl_mod.F90", "cloudsc_driver_ad_mod.F90"):
        cmd = [FC, "-cpp", "-O2", "-fPIC", "-module-dir", str(tmp_path), "-I", str(tmp_path), "-I", REFMODS, "-c",
               os.path.join(FDIR, src), "-o", str(tmp_path / (src[:-4] + ".o"))]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (" ".join(cmd), r.stderr[-3000:])
