"""The fp32 build (the reference's -DSINGLE): libcloudsc2_hip_sp.so against the reference Fortran
compiled with -DSINGLE (oracle/_ref/libcloudsc2_ref_sp.so).  A process works in ONE precision (like one binary of the
reference), so the checks run in a child process with CLOUDSC2_PRECISION=single; see tests/single_checks.py for what
is asserted and the tolerances."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import pytest

from tests.util import ROOT, refcall

SP_LIB = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "csrc", "libcloudsc2_hip_sp.so")


def _child(what: str, timeout: int):
    env = dict(os.environ, CLOUDSC2_PRECISION="single", CLOUDSC2_MATH="fast")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "single_checks.py"), what], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    print(p.stdout[-6000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "SINGLE CHECKS PASSED" in p.stdout
    return p.stdout


def test_single_library_exports_the_same_abi():
    from dwarf_p_cloudsc2_tl_ad_amd import binding as B

    assert os.path.exists(SP_LIB), "libcloudsc2_hip_sp.so not built (make -C dwarf_p_cloudsc2_tl_ad_amd/csrc libcloudsc2_hip_sp.so)"
    lib = C.CDLL(SP_LIB)
    for name in B.EXPORTED:
        assert hasattr(lib, name), name
    lib.cloudsc2_real_bytes.restype = C.c_int
    assert lib.cloudsc2_real_bytes() == 4
    assert B.lib.cloudsc2_real_bytes() == 8


@pytest.mark.skipif(not refcall.have_ref(single=True), reason="oracle/_ref/libcloudsc2_ref_sp.so not built")
def test_single_column_code_on_host():
    out = _child("host", 900)
    assert out.count("FAIL") == 0


@pytest.mark.gpu
@pytest.mark.skipif(not refcall.have_ref(single=True), reason="oracle/_ref/libcloudsc2_ref_sp.so not built")
def test_single_kernels_on_gpu():
    out = _child("gpu", 900)
    assert "driver-level NL == kernel-level NL" in out and "fp32 criterion OK" in out


@pytest.mark.gpu
def test_single_fortran_mains(tmp_path):
    """fortran/build_sp: the drivers with the reference's signatures compiled with -DSINGLE (JPRB = fp32) against
    libcloudsc2_hip_sp.so.  The NL main runs and reports; the adjoint test prints TEST FAILED exactly as the reference's own
    -DSINGLE binary does, because the verdict divides by EPSILON(1._8) whatever JPRB is (cloudsc_driver_ad_mod.F90:258-264)
    -- in units of the fp32 epsilon the error is a few eps (the fp32-only criterion of tests/single_checks_gpu.py); the
    Taylor test stops with "TL is totally wrong" exactly where the reference's -DSINGLE binary
    does (lambda = 1e-10 vanishes in fp32, cloudsc_driver_tl_mod.F90:247-249)."""
    bld = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran", "build_sp")
    if not os.path.exists(os.path.join(bld, "dwarf-cloudsc2-nl")):
        pytest.fail("fortran/build_sp missing: run __graft_entry__.build()")
    run = lambda exe, *a: subprocess.run([os.path.join(bld, exe), *map(str, a)], capture_output=True, text=True,  # noqa: E731
                                         timeout=300, cwd=tmp_path)
    r = run("dwarf-cloudsc2-nl", 4, 16000, 32)
    assert r.returncode == 0 and "NGPBLKS=500" in r.stderr, r.stdout + r.stderr
    assert "PFPLSL" in r.stdout
    r = run("dwarf-cloudsc2-ad", 1, 100, 100)
    assert r.returncode == 0 and "TEST FAILED" in r.stdout, r.stdout + r.stderr
    import re

    m = re.search(r"maximum error is\s+([0-9.Ee+-]+)", r.stdout)
    assert m, r.stdout
    assert float(m.group(1)) * (2.220446049250313e-16 / 1.1920928955078125e-07) < 1e4, r.stdout  # a few fp32 epsilons
    r = run("dwarf-cloudsc2-tl", 1, 100, 1)
    assert "TL is totally wrong" in (r.stdout + r.stderr), r.stdout + r.stderr
