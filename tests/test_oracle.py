"""CPU tests of the checker itself: the C restatement (oracle/cloudsc2_oracle.c) against
  (a) golden vectors generated from the unmodified reference Fortran (tests/golden/, always available), and
  (b) the reference Fortran itself (oracle/_ref), when it has been built in this container.
"""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from tests.golden.make_golden import input_digest, table_inputs, tl_ad
from tests.util import ROOT, c2, make_params, refcall, relerr, set_lib_params

GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "drivers.json")))

# the restatement follows the reference statement by statement and is compiled without FMA contraction like the
# flang build of the reference: agreement is expected to the last bit; 1e-13 leaves room for another libm.
ORACLE_TOL = 1e-13


@pytest.fixture(scope="module")
def oracle():
    if not refcall.have_oracle():
        pytest.fail("oracle/libcloudsc2_oracle.so missing: run `make -C oracle oracle` (or __graft_entry__.build())")
    return refcall.OracleLib()


def _table(name):
    info = META["fixtures"][name]
    if name.startswith("evap"):
        tab = c2.random_table(137, info["ncol"], seed=info["seed"])
    else:
        tab = c2.synthetic_table()
        tab = {k: (v[:, : info["ncol"]] if isinstance(v, np.ndarray) else v) for k, v in tab.items()}
    return tab, info


def test_golden_nl(oracle):
    tab, info = _table("nl_synth100")
    g = np.load(os.path.join(GOLD, "nl_synth100.npz"))
    set_lib_params(oracle, make_params(tab))
    st, inp = table_inputs(oracle, tab, info["ncol"])
    assert relerr(g["qsat"], inp["qsat"]) <= ORACLE_TOL
    assert input_digest(inp) == info["digest"], "synthetic inputs or SATUR drifted from the fixture"
    out = oracle.cloudsc2(st.ptsphy, inp)
    for n, a in out.items():
        assert relerr(g["out_" + n], a) <= ORACLE_TOL, n
    # the golden data precipitates in every column (the reference's Taylor test needs that) and has rain and snow
    assert np.all(np.abs(g["out_fplsn"][-1]) + np.abs(g["out_fplsl"][-1]) > 0)
    assert np.any(g["out_fplsl"] > 0) and np.any(g["out_fplsn"] > 0)


@pytest.mark.parametrize("lreg", [0, 1])
def test_golden_tl_ad(oracle, lreg):
    name = f"tlad_synth24_r{lreg}"
    tab, info = _table(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    set_lib_params(oracle, make_params(tab, lregcl=bool(lreg)))
    st, inp = table_inputs(oracle, tab, info["ncol"])
    assert input_digest(inp) == info["digest"]
    out5, dout, x = tl_ad(oracle, st.ptsphy, inp)
    for n in dout:
        assert relerr(g["tl_" + n], dout[n]) <= ORACLE_TOL, ("tl", n)
        assert relerr(g["traj_" + n], out5[n]) <= ORACLE_TOL, ("traj", n)
    for n in x:
        assert relerr(g["ad_" + n], x[n]) <= ORACLE_TOL, ("ad", n)


def test_golden_evaporation_branch(oracle):
    tab, info = _table("evap_rand24")
    g = np.load(os.path.join(GOLD, "evap_rand24.npz"))
    set_lib_params(oracle, make_params(tab, **info["flags"]))
    st, inp = table_inputs(oracle, tab, info["ncol"])
    assert input_digest(inp) == info["digest"]
    out = oracle.cloudsc2(st.ptsphy, inp)
    out5, dout, x = tl_ad(oracle, st.ptsphy, inp)
    assert np.any(g["out_covptot"] != 0.0)
    for n in out:
        assert relerr(g["out_" + n], out[n]) <= ORACLE_TOL, n
        assert relerr(g["tl_" + n], dout[n]) <= ORACLE_TOL, ("tl", n)
    for n in x:
        assert relerr(g["ad_" + n], x[n]) <= ORACLE_TOL, ("ad", n)


def test_reference_h5_constants():
    """config-files/reference.h5 (the reference's golden NL output) cannot be reproduced without the undistributed
    input.h5, but it pins two things: RLSTT (-PFHPSN/PFPLSN, cloudsc2.F90:733) and the output conventions
    (PCOVPTOT, TENDENCY_LOC_A, rain/snow species of CLD never written)."""
    g = np.load(os.path.join(GOLD, "reference_h5.npz"))
    prm = c2.default_params()
    nz = g["PFPLSN"] != 0
    assert nz.sum() == 4497
    assert np.max(np.abs(-g["PFHPSN"][nz] / g["PFPLSN"][nz] / prm.rlstt - 1.0)) < 1e-15
    assert not g["PCOVPTOT"].any() and not g["TENDENCY_LOC_A"].any() and not g["PFPLSL"].any()
    assert not g["TENDENCY_LOC_CLD"][2:].any() and g["TENDENCY_LOC_CLD"][:2].any()
    assert abs(np.abs(g["TENDENCY_LOC_T"]).sum() - 8.280448058210742e-02) < 1e-15


@pytest.mark.skipif(not refcall.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("flags", [dict(), dict(lregcl=True), dict(levapls2=True), dict(levapls2=True, lregcl=True)])
def test_oracle_equals_reference(oracle, flags):
    tab = c2.random_table(137, 48, seed=21)
    prm = make_params(tab, **flags)
    ref = refcall.RefLib()
    set_lib_params(ref, prm)
    set_lib_params(oracle, prm)
    st, inp = table_inputs(ref, tab, 48)
    assert np.array_equal(inp["qsat"], oracle.satur(inp["pap"], inp["t"]))
    a, b = ref.cloudsc2(st.ptsphy, inp), oracle.cloudsc2(st.ptsphy, inp)
    for n in a:
        assert relerr(a[n], b[n]) <= ORACLE_TOL, ("nl", n)
    ra, rb = tl_ad(ref, st.ptsphy, inp), tl_ad(oracle, st.ptsphy, inp)
    for k in range(3):
        for n in ra[k]:
            assert relerr(ra[k][n], rb[k][n]) <= ORACLE_TOL, (k, n)


@pytest.mark.skipif(not refcall.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_reference_without_lphylin(oracle):
    """YREPHLI%LPHYLIN=.false. (cloudsc2.F90:365-369, FOEALFA / FOEEWM): the NL restatement against the reference with the switch
    off -- bit for bit like the shipped configuration -- and the switch changes the results."""
    tab = c2.random_table(137, 48, seed=23)
    prm = make_params(tab)
    ref = refcall.RefLib()
    set_lib_params(ref, prm)
    st, inp = table_inputs(ref, tab, 48)
    lin = ref.cloudsc2(st.ptsphy, inp)
    prm.lphylin = 0
    set_lib_params(ref, prm)
    set_lib_params(oracle, prm)
    a, b = ref.cloudsc2(st.ptsphy, inp), oracle.cloudsc2(st.ptsphy, inp)
    for n in a:
        assert relerr(a[n], b[n]) <= ORACLE_TOL, ("nl", n)
    assert not np.array_equal(a["tent"], lin["tent"])
    prm.lphylin = 1
    set_lib_params(ref, prm)  # the reference library's module state is process-wide: leave it as the other tests expect it
    set_lib_params(oracle, prm)


@pytest.mark.skipif(not refcall.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_driver_equals_reference_driver(oracle):
    """The OpenMP block loop used for the CPU baseline when the reference build cannot travel."""
    import ctypes as C

    tab = c2.synthetic_table()
    prm = make_params(tab)
    ref = refcall.RefLib()
    set_lib_params(ref, prm)
    set_lib_params(oracle, prm)
    a = c2.state_from_table(tab, 32, 230)
    b = a.copy()
    ref.driver(0, 2, 32, a.nlev, 230, a.ptsphy, a.driver_arrays())
    dp = C.POINTER(C.c_double)
    oracle.lib.oracle_driver_nl.argtypes = [C.c_int] * 4 + [C.c_double] + [dp] * 18
    oracle.lib.oracle_driver_nl(2, 32, b.nlev, 230, b.ptsphy, *[x.ctypes.data_as(dp) for x in b.driver_arrays()])
    for n in a.outputs():
        assert np.array_equal(a.outputs()[n], b.outputs()[n]), n


PY_REF = "/root/reference/src/cloudsc2_nl_gt4py/cloudsc2_py.py"


@pytest.mark.skipif(not os.path.exists(PY_REF), reason="the reference checkout (its numpy restatement of SATUR + CLOUDSC2) is not here")
def test_oracle_equals_the_references_python_restatement(oracle):
    """A third, independent pin of the oracle: the reference's own numpy restatement of SATUR + CLOUDSC2
    (src/cloudsc2_nl_gt4py/cloudsc2_py.py, imported from the read-only checkout, numpy + math only, LEVAPLS2 hard-wired
    off) on synthetic columns -- agreement to rounding with the C restatement the GPU path is checked against."""
    import importlib.util
    import types

    spec = importlib.util.spec_from_file_location("_ref_cloudsc2_py", PY_REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    tab = c2.synthetic_table()
    ncol = 24
    tab = {k: (np.ascontiguousarray(v[:, :ncol]) if isinstance(v, np.ndarray) else v) for k, v in tab.items()}
    prm = make_params(tab)
    set_lib_params(oracle, prm)
    st = c2.state_from_table(tab, ncol, ncol, real=np.float64)
    nlev = st.nlev
    ns = types.SimpleNamespace
    cst = ns(rg=prm.rg, rd=prm.rd, rcpd=prm.rcpd, retv=prm.retv, rlvtt=prm.rlvtt, rlstt=prm.rlstt, rlmlt=prm.rlmlt, rtt=prm.rtt)
    ethf = ns(r2es=prm.r2es, r3les=prm.r3les, r3ies=prm.r3ies, r4les=prm.r4les, r4ies=prm.r4ies, r5les=prm.r5les,
              r5ies=prm.r5ies, r5alvcp=prm.r5alvcp, r5alscp=prm.r5alscp, ralvdcp=prm.ralvdcp, ralsdcp=prm.ralsdcp,
              rtwat=prm.rtwat, rtice=prm.rtice, rtwat_rtice_r=prm.rtwat_rtice_r, rvtmp2=prm.rvtmp2)
    ecldp = ns(rclcrit=prm.rclcrit, rkconv=prm.rkconv, rlmin=prm.rlmin, rpecons=prm.rpecons)
    ecld = ns(ceta=prm.ceta_array())
    ephli = ns(lphylin=True, rlptrc=prm.rlptrc)

    inp = refcall.block_inputs(st, 0)
    qs_py = np.zeros((nlev, ncol))
    mod.satur(1, ncol, ncol, 1, nlev, True, inp["pap"], inp["t"], qs_py, 2, ethf, cst)
    qs = oracle.satur(inp["pap"], inp["t"])
    assert relerr(qs, qs_py) <= ORACLE_TOL
    inp["qsat"] = qs
    want = oracle.cloudsc2(st.ptsphy, inp)
    out = refcall.new_outputs(nlev, ncol)
    a = refcall.kernel_arg_order({k: v.copy() for k, v in inp.items()}, out)
    mod.cloudsc2_py(1, ncol, ncol, 1, nlev, False, st.ptsphy, *a, ecldp, ecld, cst, ethf, ephli)
    for n in want:
        assert relerr(want[n], out[n]) <= 5e-13, n
