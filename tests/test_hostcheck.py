"""The per-column device functions (csrc/cloudsc2_level.hpp + cloudsc2_column.hpp) compiled for the HOST and run
against the oracle -- exercises exactly the code the HIP kernels wrap (level physics, lane indexing, prefetch
double-buffer, checkpoint/recompute adjoint sweep) in a container without a GPU.  The host build is a unit-test
vehicle only; the real parity tests are tests/test_gpu_parity.py through the C ABI on the MI355X.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from tests.util import (B, c2, flat_block, flat_fields, host_traj_blocks, hostcheck, increments_of, make_params, refcall,
                        relerr, set_lib_params)

TOL = 1e-12  # host build: no FMA contraction, host libm -> differences only from the re-associated adjoint sums


@pytest.fixture(scope="module")
def oracle():
    return refcall.OracleLib()


@pytest.fixture(params=["fast", "precise"], autouse=True)
def math_mode(request):
    """Both arithmetic variants of the kernels: shared-reciprocal fast math (default) and the reference's own
    operation order with IEEE division / libm exp (what the Taylor-test driver uses)."""
    hostcheck().hostcheck_set_precise(int(request.param == "precise"))
    yield request.param
    hostcheck().hostcheck_set_precise(0)


def oracle_qsat(oracle, st):
    q = np.zeros_like(st.PAP)
    for ibl in range(st.nblocks):
        icend = min(st.nproma, st.ngptot - ibl * st.nproma)
        q[ibl] = oracle.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        q[ibl][:, icend:] = 0.0
    return q


@pytest.mark.parametrize("nproma,ngptot", [(32, 100), (100, 100), (1, 5), (48, 130)])
def test_nl_columns(oracle, nproma, ngptot):
    tab = c2.synthetic_table()
    prm = make_params(tab)
    set_lib_params(oracle, prm)
    st = c2.state_from_table(tab, nproma, ngptot, poison_outputs=-5.0)
    got = st.copy()
    i, o = host_traj_blocks(got)
    S = nproma * st.nlev
    zero = B.Field()
    zero.ptr = got.B_LOC.ctypes.data + 8 * 7 * S
    zero.block_stride = 8 * S
    assert hostcheck().hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i), C.byref(o), zero, 0.0) == 0
    # the 32-bit byte-offset variant of the sweep (C2F_OFF32) must give the same bits
    got32 = st.copy()
    i32, o32 = host_traj_blocks(got32)
    zero32 = B.Field()
    zero32.ptr = got32.B_LOC.ctypes.data + 8 * 7 * S
    zero32.block_stride = 8 * S
    hostcheck().hostcheck_set_off32(1)
    try:
        assert hostcheck().hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i32), C.byref(o32), zero32, 0.0) == 0
    finally:
        hostcheck().hostcheck_set_off32(0)
    for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN"):
        assert np.array_equal(getattr(got, n), getattr(got32, n)), n
    for ibl in range(st.nblocks):
        icend = min(nproma, ngptot - ibl * nproma)
        qs = oracle.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        inp = refcall.block_inputs(st, ibl, qs)
        for a in inp.values():
            a[:, icend:] = 1.0
        want = oracle.cloudsc2(st.ptsphy, inp, kfdia=icend)
        for n, a in refcall.state_outputs_block(got, ibl).items():
            assert relerr(want[n][:, :icend], a[:, :icend]) <= TOL, (n, ibl)
            if n != "covptot":
                assert np.all(a[:, icend:] == -5.0), ("tail touched", n)
    assert np.all(got.PCOVPTOT == 0.0) and np.all(got.B_LOC[:, 7] == 0.0)


@pytest.mark.parametrize("flags,nlev", [(dict(), 137), (dict(lregcl=True), 137), (dict(levapls2=True, lregcl=True), 137),
                                        (dict(ldrain1d=True), 137), (dict(), 60), (dict(levapls2=True, lregcl=True), 200),
                                        (dict(), 11), (dict(), 2), (dict(), 1)])
def test_tl_ad_columns(oracle, flags, nlev):
    """NLEV other than 137: the level tables, the tropopause band (empty for very few levels) and CETA(200) bounds."""
    tab = c2.random_table(nlev, 30, seed=4)
    prm = make_params(tab, **flags)
    set_lib_params(oracle, prm)
    nproma, ngptot = 16, 30
    st = c2.state_from_table(tab, nproma, ngptot)
    nb, nlev = st.nblocks, st.nlev
    qsat = oracle_qsat(oracle, st)
    inc = increments_of(st, qsat)
    hc = hostcheck()

    got = st.copy()
    i, o = host_traj_blocks(got, qsat)
    tl = flat_fields("out", nb, nlev, nproma)
    di, do_ = flat_block("in", inc), flat_block("out", tl)
    assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i), C.byref(o), C.byref(di), C.byref(do_)) == 0

    x = flat_fields("in", nb, nlev, nproma)
    y = {n: a.copy() for n, a in tl.items()}
    scratch = np.zeros((nb, nlev, nproma))
    got2 = st.copy()
    i2, o2 = host_traj_blocks(got2, qsat)
    ai, ao = flat_block("in", x), flat_block("out", y)
    assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i2), C.byref(o2), C.byref(ai), C.byref(ao),
                           scratch.ctypes.data) == 0

    # the 32-bit byte-offset variants of the TL and AD sweeps (C2F_OFF32) give the same bits, and the checkpoint stores
    # stay inside the scratch plane (guard block behind it)
    hc.hostcheck_set_off32(1)
    try:
        got3 = st.copy()
        i3, o3 = host_traj_blocks(got3, qsat)
        tl3 = flat_fields("out", nb, nlev, nproma)
        di3, do3 = flat_block("in", inc), flat_block("out", tl3)
        assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i3), C.byref(o3), C.byref(di3), C.byref(do3)) == 0
        x3 = flat_fields("in", nb, nlev, nproma)
        y3 = {n: a.copy() for n, a in tl3.items()}
        guard = np.full((nb + 1, nlev, nproma), 7.0)
        ai3, ao3 = flat_block("in", x3), flat_block("out", y3)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i3), C.byref(o3), C.byref(ai3), C.byref(ao3),
                               guard.ctypes.data) == 0
    finally:
        hc.hostcheck_set_off32(0)
    assert np.all(guard[nb] == 7.0)
    for n in tl:
        assert np.array_equal(tl[n], tl3[n]), ("off32 tl", n)
    for n in x:
        assert np.array_equal(x[n], x3[n]), ("off32 ad", n)

    # the self-increment form of the TL sweep (C2F_SELFINC, cloudsc2_tl_launch_self: dx = 0.01*x formed in the sweep, as both test
    # drivers define it) equals the sweep fed with those increments from memory -- same operations, no contraction on this host build
    for sup in (0.01, 0.0):
        hc.hostcheck_set_self_increment(sup)
        try:
            got5 = st.copy()
            i5, o5 = host_traj_blocks(got5, qsat)
            tl5 = flat_fields("out", nb, nlev, nproma)
            do5 = flat_block("out", tl5)
            assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i5), C.byref(o5), None, C.byref(do5)) == 0
        finally:
            hc.hostcheck_set_self_increment(-1.0)
        if sup == 0.0:
            inc0 = increments_of(st, qsat, zero_supsat=True)
            got6 = st.copy()
            i6, o6 = host_traj_blocks(got6, qsat)
            tl6 = flat_fields("out", nb, nlev, nproma)
            di6, do6 = flat_block("in", inc0), flat_block("out", tl6)
            assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i6), C.byref(o6), C.byref(di6), C.byref(do6)) == 0
            want = tl6
        else:
            want = tl
        for n in tl:
            assert np.array_equal(want[n], tl5[n]), ("self-increment tl", sup, n)

    # the assign form of the adjoint (C2F_ASSIGN, cloudsc2_ad_launch_assign): x = A^T y into arrays full of garbage must equal
    # the accumulate form into zeroed arrays bit for bit on the active columns, and must not touch the padded tail
    hc.hostcheck_set_assign(1)
    try:
        got4 = st.copy()
        i4, o4 = host_traj_blocks(got4, qsat)
        x4 = {n: np.full_like(a, 7.25) for n, a in x.items()}
        y4 = {n: a.copy() for n, a in tl.items()}
        ai4, ao4 = flat_block("in", x4), flat_block("out", y4)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i4), C.byref(o4), C.byref(ai4), C.byref(ao4),
                               scratch.ctypes.data) == 0
    finally:
        hc.hostcheck_set_assign(0)
    for ibl in range(nb):
        icend = min(nproma, ngptot - ibl * nproma)
        for n in x:
            assert np.array_equal(x[n][ibl][:, :icend], x4[n][ibl][:, :icend]), ("assign", n)
            assert np.all(x4[n][ibl][:, icend:] == 7.25), ("assign touched the tail", n)
        for n in y4:
            assert np.all(y4[n][ibl][:, :icend] == 0.0), ("assign: output adjoint not consumed", n)

    # The reverse sweep alone (cloudsc2_ad_launch_reverse) on the trajectory outputs the TL sweep left behind equals the whole
    # adjoint bit for bit; it reads nothing of the trajectory outputs but PFPLSL5 / PFPLSN5 (everything else is poisoned here) and,
    # without the evaporation branch, neither sweep touches the cover-checkpoint plane (NaN-filled: a read would poison x).
    evap = bool(flags.get("levapls2", False) or flags.get("ldrain1d", False))
    hc.hostcheck_set_ad_sweep(2)
    try:
        got5 = got.copy()  # the state after hostcheck_tl: trajectory outputs included
        for n in ("B_LOC", "PA", "PCOVPTOT", "PFHPSL", "PFHPSN"):
            getattr(got5, n)[...] = np.nan
        i5, o5_ = host_traj_blocks(got5, qsat)
        x5 = flat_fields("in", nb, nlev, nproma)
        y5 = {n: a.copy() for n, a in tl.items()}
        ck5 = scratch.copy() if evap else np.full((nb, nlev, nproma), np.nan)
        ai5, ao5 = flat_block("in", x5), flat_block("out", y5)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i5), C.byref(o5_), C.byref(ai5), C.byref(ao5),
                               ck5.ctypes.data) == 0
    finally:
        hc.hostcheck_set_ad_sweep(0)
    for n in x:
        assert np.array_equal(x[n], x5[n]), ("reverse sweep alone", n)
    for n in y5:
        assert np.array_equal(y[n], y5[n]), ("reverse sweep alone: output adjoints", n)
    if not evap:
        assert np.all(scratch == 0.0), "the forward sweep wrote cover checkpoints nobody reads"
        assert np.all(np.isnan(ck5))

    ld = bool(flags.get("ldrain1d", False))
    for ibl in range(nb):
        icend = min(nproma, ngptot - ibl * nproma)
        inp = refcall.block_inputs(st, ibl, qsat[ibl])
        dinp = {n: np.ascontiguousarray(inc[n][ibl]) for n in inc}
        for d in (inp, dinp):
            for a in d.values():
                a[:, icend:] = 1.0
        o5, dout = oracle.cloudsc2tl(st.ptsphy, inp, dinp, kfdia=icend, ldrain1d=ld)
        for n in dout:
            assert relerr(dout[n][:, :icend], tl[n][ibl][:, :icend]) <= TOL, ("tl", n)
        for n, a in refcall.state_outputs_block(got, ibl).items():
            assert relerr(o5[n][:, :icend], a[:, :icend]) <= TOL, ("traj", n)
        xr = refcall.new_inputs(nlev, nproma)
        yr = {n: a.copy() for n, a in dout.items()}
        oracle.cloudsc2ad(st.ptsphy, inp, xr, yr, kfdia=icend, ldrain1d=ld)
        for n in xr:
            assert relerr(xr[n][:, :icend], x[n][ibl][:, :icend]) <= TOL, ("ad", n)
            assert np.all(x[n][ibl][:, icend:] == 0.0)
        for n in y:
            assert np.all(y[n][ibl][:, :icend] == 0.0), ("output adjoint not consumed", n)

    # <TL x, TL x> = <x, AD TL x> per column (PSUPSAT excluded: the reference assigns its adjoint, cloudsc2ad.F90:1733);
    # with one or two levels the columns are near-trivial and the relative form of the identity is ill-conditioned
    #  -- parity with the checker above still holds there
    for ibl in range(nb if nlev >= 10 else 0):
        icend = min(nproma, ngptot - ibl * nproma)
        n1 = sum((tl[n][ibl][:, :icend] ** 2).sum(axis=0) for n in tl)
        n2 = sum((inc[n][ibl][:, :icend] * x[n][ibl][:, :icend]).sum(axis=0) for n in inc if n != "supsat")
        n2s = (inc["supsat"][ibl][:, :icend] * x["supsat"][ibl][:, :icend] / st.ptsphy).sum(axis=0)  # corrected term
        assert np.max(np.abs(n1 - n2 - n2s) / np.abs(n1)) < 1e4 * 2.2e-16


@pytest.mark.parametrize("ldrain1d", [False, True])
def test_nl_without_lphylin(oracle, ldrain1d):
    """YREPHLI%LPHYLIN=.false. (off in every reference main): the FOEALFA / FOEEWM form of stage A (cloudsc2.F90:365-369), unless
    LDRAIN1D switches the linearised form back on (:349)."""
    tab = c2.random_table(137, 40, seed=9)
    prm = make_params(tab, ldrain1d=ldrain1d)
    prm.lphylin = 0
    set_lib_params(oracle, prm)
    nproma, ngptot = 16, 40
    st = c2.state_from_table(tab, nproma, ngptot)
    got = st.copy()
    i, o = host_traj_blocks(got)
    assert hostcheck().hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i), C.byref(o), B.Field(), 0.0) == 0
    lin = st.copy()
    prm_lin = make_params(tab, ldrain1d=ldrain1d)
    il, ol = host_traj_blocks(lin)
    assert hostcheck().hostcheck_nl(C.byref(prm_lin), st.ptsphy, nproma, st.nlev, ngptot, C.byref(il), C.byref(ol), B.Field(), 0.0) == 0
    differs = False
    for ibl in range(st.nblocks):
        icend = min(nproma, ngptot - ibl * nproma)
        qs = oracle.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        inp = refcall.block_inputs(st, ibl, qs)
        for a in inp.values():
            a[:, icend:] = 1.0
        want = oracle.cloudsc2(st.ptsphy, inp, kfdia=icend, ldrain1d=ldrain1d)
        for n, a in refcall.state_outputs_block(got, ibl).items():
            assert relerr(want[n][:, :icend], a[:, :icend]) <= TOL, (n, ibl)
            differs |= not np.array_equal(a[:, :icend], refcall.state_outputs_block(lin, ibl)[n][:, :icend])
    assert differs != ldrain1d  # the switch changes the results -- except under LDRAIN1D, where it is not read


def test_perturbed_nl_matches_explicit_perturbation(oracle):
    """pert_lambda of cloudsc2_nl_launch = the Taylor test's x + lambda*(0.01 x) (cloudsc_driver_tl_mod.F90:200-215)."""
    tab = c2.synthetic_table()
    prm = make_params(tab)
    set_lib_params(oracle, prm)
    nproma = ngptot = 20
    st = c2.state_from_table(tab, nproma, ngptot)
    qsat = oracle_qsat(oracle, st)
    lam = 1e-3
    got = st.copy()
    i, o = host_traj_blocks(got, qsat)
    hostcheck().hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i), C.byref(o), B.Field(), lam)
    inp = refcall.block_inputs(st, 0, qsat[0])
    pin = {n: a + lam * (a * 0.01) for n, a in inp.items()}
    want = oracle.cloudsc2(st.ptsphy, pin)
    for n, a in refcall.state_outputs_block(got, 0).items():
        assert relerr(want[n], a) <= TOL, n


@pytest.mark.parametrize("nproma, ngptot, precise, off32, levapls2", [(8, 14, 0, 0, False), (5, 23, 1, 1, False), (16, 16, 0, 1, True)])
def test_lambda_sweep_of_the_taylor_test(oracle, nproma, ngptot, precise, off32, levapls2):
    """taylor_column (the ten lambdas of cloudsc_driver_tl_mod.F90:197-244 on the lanes of a wave: lane (c, k) of wave w runs
    column 6w + c perturbed by 10^-(k+1), compares with the stored base run level by level and keeps the level sums) against ten
    explicit perturbed NL sweeps of the same build: the terms are the same operations, the level sums are taken in the same order."""
    tab = c2.synthetic_table()
    prm = make_params(tab, levapls2=levapls2)
    set_lib_params(oracle, prm)
    hc = hostcheck()
    hc.hostcheck_set_precise(precise)
    hc.hostcheck_set_off32(off32)
    try:
        st = c2.state_from_table(tab, nproma, ngptot)
        qsat = oracle_qsat(oracle, st)
        base = st.copy()
        i, o = host_traj_blocks(base, qsat)
        hc.hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i), C.byref(o), B.Field(), 0.0)
        rng = np.random.default_rng(5)
        tl = flat_fields("out", st.nblocks, st.nlev, nproma)
        for a in tl.values():
            a[...] = rng.standard_normal(a.shape)
        for n in ("fplsl", "fplsn", "fhpsl", "fhpsn"):
            tl[n][:, 0, :] = 0.0  # the fluxes' top value is zero in every run (cloudsc2tl.F90: PFPLSL(JL,1) = 0)
        tlb = flat_block("out", tl)
        ncols_pad = st.nblocks * nproma
        colsum = np.full((110, ncols_pad), np.nan)
        hc.hostcheck_taylor_sweep(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(i), C.byref(o), C.byref(tlb),
                                  colsum.ctypes.data_as(C.c_void_p))
        names = ("tent", "tenq", "tenl", "teni", "clc", "fplsl", "fplsn", "fhpsl", "fhpsn", "covptot")  # order of the ERROR_NORM calls

        def columns(arrs):  # name -> (levels, all columns) with the fluxes on the half levels below (what a level of the sweep sees)
            out = {}
            for n in names:
                a = arrs[n]
                a = a[:, 1:, :] if a.shape[1] == st.nlev + 1 else a
                out[n] = np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(st.nlev, ncols_pad)
            return out

        def out_arrays(s):
            return {"tent": s.B_LOC[:, 0], "tenq": s.B_LOC[:, 2], "tenl": s.B_LOC[:, 3], "teni": s.B_LOC[:, 4], "clc": s.PA,
                    "covptot": s.PCOVPTOT, "fplsl": s.PFPLSL, "fplsn": s.PFPLSN, "fhpsl": s.PFHPSL, "fhpsn": s.PFHPSN}

        b = columns(out_arrays(base))
        act = np.arange(ncols_pad) < ngptot  # (blocks are contiguous: global column = ibl*nproma + jl)
        assert np.all(np.isnan(colsum[:, ~act])) and np.all(np.isfinite(colsum[:, act]))
        tlc = columns(tl)
        for f, n in enumerate(names):
            acc = np.zeros(ncols_pad)
            for jk in range(st.nlev):
                acc += tlc[n][jk].astype(np.float64)
            assert np.array_equal(colsum[100 + f, act], acc[act]), n
        for k in range(10):
            lam = 10.0 ** -(k + 1)
            pert = st.copy()
            ip, op = host_traj_blocks(pert, qsat)
            hc.hostcheck_nl(C.byref(prm), st.ptsphy, nproma, st.nlev, ngptot, C.byref(ip), C.byref(op), B.Field(), lam)
            pc = columns(out_arrays(pert))
            for f, n in enumerate(names):
                acc = np.zeros(ncols_pad)
                for jk in range(st.nlev):
                    acc += (b[n][jk] - pc[n][jk]).astype(np.float64)
                got = colsum[k * 10 + f, act]
                scale = np.abs(b[n]).max() * st.nlev
                assert np.all(np.abs(got - acc[act]) <= 4 * np.finfo(B.REAL).eps * scale), (k, n, np.abs(got - acc[act]).max(), scale)
    finally:
        hc.hostcheck_set_precise(0)
        hc.hostcheck_set_off32(0)


def test_adjoint_test_norms_formed_in_the_sweeps(oracle):
    """The adjoint test's three norms (cloudsc_driver_ad_mod.F90:184-195,240-264) as the library's driver forms them: <y,y> in the
    TL sweep that produces y (cloudsc2_tl_launch_self's yy), <x0,x_adj> and norm3 in the reverse sweep that produces x_adj
    (cloudsc2_ad_launch_reverse_norms) -- against the sums over the stored arrays."""
    tab = c2.synthetic_table()
    prm = make_params(tab, lregcl=True)
    set_lib_params(oracle, prm)
    nproma, ngptot = 16, 40
    st = c2.state_from_table(tab, nproma, ngptot)
    nb, nlev = st.nblocks, st.nlev
    qsat = oracle_qsat(oracle, st)
    hc = hostcheck()
    ncols_pad = nb * nproma
    norms = np.full((3, ncols_pad), np.nan)
    got = st.copy()
    i, o = host_traj_blocks(got, qsat)
    y = flat_fields("out", nb, nlev, nproma)
    x = {n: np.full_like(a, 3.5) for n, a in flat_fields("in", nb, nlev, nproma).items()}
    try:
        hc.hostcheck_set_self_increment(0.0)
        hc.hostcheck_set_yy(norms[0].ctypes.data_as(C.c_void_p))
        yb = flat_block("out", y)
        assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i), C.byref(o), None, C.byref(yb)) == 0
        ysaved = {n: a.copy() for n, a in y.items()}
        hc.hostcheck_set_ad_sweep(2)
        hc.hostcheck_set_assign(1)
        hc.hostcheck_set_ad_norms(norms.ctypes.data_as(C.c_void_p))
        xb = flat_block("in", x)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i), C.byref(o), C.byref(xb), C.byref(yb), None) == 0
        n3max = hc.hostcheck_get_norm3_max()
    finally:
        hc.hostcheck_set_self_increment(-1.0)
        hc.hostcheck_set_yy(None)
        hc.hostcheck_set_ad_sweep(0)
        hc.hostcheck_set_assign(0)
        hc.hostcheck_set_ad_norms(None)
    act = (np.arange(ncols_pad) < ngptot)

    def cols(a):  # (blocks, levels, nproma) -> (levels, all columns)
        return np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(a.shape[1], ncols_pad).astype(np.float64)

    n1 = sum((cols(a) ** 2).sum(axis=0) for a in ysaved.values())
    assert np.all(np.isnan(norms[:, ~act]))
    assert np.allclose(norms[0, act], n1[act], rtol=1e-13, atol=0)
    x0 = increments_of(st, qsat, zero_supsat=True)
    n2 = sum((cols(x0[n]) * cols(x[n])).sum(axis=0) for n in x if n != "supsat")
    assert np.allclose(norms[1, act], n2[act], rtol=1e-12, atol=0), np.abs(norms[1, act] / n2[act] - 1).max()
    eps = np.finfo(np.float64).eps
    n3 = np.abs(norms[0] - norms[1]) / eps / norms[1]
    assert np.array_equal(norms[2, act], n3[act])
    assert n3max == np.abs(n3[act]).max()
    assert n3max < 1e4  # the adjoint identity itself (cloudsc_driver_ad_mod.F90:289)
