"""fp32 (-DSINGLE) checks, run as a CHILD process with CLOUDSC2_PRECISION=single by tests/test_single.py:
``python tests/single_checks.py host|gpu``.  One precision per process, like one -DSINGLE binary of the reference.

Checker: oracle/_ref/libcloudsc2_ref_sp.so = the unmodified reference Fortran built with its own -DSINGLE
(parkind1.F90:40-41).  No golden data exists for fp32, so three things are asserted:
  1. against the fp32 reference, per output field, max-norm relative difference <= TOL_VS_REF32;
  2. against the fp64 reference on the same (fp32-rounded) inputs, our fp32 error is at most ERR_FACTOR x the error
     the reference's own fp32 build makes, plus a floor -- "as accurate as the reference's -DSINGLE";
  3. <TL dx, TL dx> = <dx, AD TL dx> per column to fp32 round-off (the reference's adjoint test, in fp32).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

os.environ["CLOUDSC2_PRECISION"] = "single"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tests.util import (B, c2, flat_block, flat_fields, host_traj_blocks, hostcheck, increments_of, make_params,  # noqa: E402
                        refcall, relerr, set_lib_params)

assert B.SINGLE and B.REAL is np.float32 and B.lib.cloudsc2_real_bytes() == 4

TOL_VS_REF32 = 2e-3   # two fp32 evaluations of the same column differ by amplified round-off (tendencies are differences / PTSPHY)
ERR_FACTOR, ERR_FLOOR = 4.0, 2e-5
F32 = np.float32


def refs(prm):
    r32, r64 = refcall.RefLib(single=True), refcall.RefLib()
    set_lib_params(r32, prm)
    set_lib_params(r64, prm)
    return r32, r64


def up(d):  # fp32 block dict -> fp64 copies (same values)
    return {n: np.ascontiguousarray(a, dtype=np.float64) for n, a in d.items()}


def qsat_blocks(r32, st):
    q = np.zeros_like(st.PAP)
    for ibl in range(st.nblocks):
        icend = min(st.nproma, st.ngptot - ibl * st.nproma)
        q[ibl] = r32.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        q[ibl][:, icend:] = 0.0
    return q


def check_field(tag, n, ours, ref32, ref64):
    """ours/ref32 fp32, ref64 fp64, all over active columns."""
    e_vs32 = relerr(ref32.astype(np.float64), ours.astype(np.float64))
    e_ours = relerr(ref64, ours.astype(np.float64))
    e_ref = relerr(ref64, ref32.astype(np.float64))
    ok = e_vs32 <= TOL_VS_REF32 and e_ours <= ERR_FACTOR * e_ref + ERR_FLOOR
    print(f"  {tag:5s} {n:8s} vs ref32 {e_vs32:9.2e}   |ours-ref64| {e_ours:9.2e}   |ref32-ref64| {e_ref:9.2e}  {'ok' if ok else 'FAIL'}")
    return ok


def compare_blocks(st, got, tl, x, inc, qsat, r32, r64, ld=False):
    """NL trajectory outputs in `got`, TL outputs `tl`, AD input adjoints `x` (after AD applied to y = TL outputs)."""
    ok = True
    nproma, nlev = st.nproma, st.nlev
    for ibl in range(st.nblocks):
        icend = min(nproma, st.ngptot - ibl * nproma)
        inp = refcall.block_inputs(st, ibl, qsat[ibl])
        dinp = {n: np.ascontiguousarray(inc[n][ibl]) for n in inc}
        for d in (inp, dinp):
            for a in d.values():
                a[:, icend:] = 1.0
        o32, d32 = r32.cloudsc2tl(st.ptsphy, inp, dinp, kfdia=icend, ldrain1d=ld)
        o64, d64 = r64.cloudsc2tl(st.ptsphy, up(inp), up(dinp), kfdia=icend, ldrain1d=ld)
        for n, a in refcall.state_outputs_block(got, ibl).items():
            ok &= check_field("traj", n, a[:, :icend], o32[n][:, :icend], o64[n][:, :icend])
        for n in d32:
            ok &= check_field("tl", n, tl[n][ibl][:, :icend], d32[n][:, :icend], d64[n][:, :icend])
        # adjoint of the SAME y in both checkers: y = the fp32 reference's TL outputs
        xr32 = refcall.new_inputs(nlev, nproma, dtype=F32)
        y32 = {n: a.copy() for n, a in d32.items()}
        r32.cloudsc2ad(st.ptsphy, inp, xr32, y32, kfdia=icend, ldrain1d=ld)
        xr64 = refcall.new_inputs(nlev, nproma)
        r64.cloudsc2ad(st.ptsphy, up(inp), xr64, up(d32), kfdia=icend, ldrain1d=ld)
        for n in xr32:
            ok &= check_field("ad", n, x[n][ibl][:, :icend], xr32[n][:, :icend], xr64[n][:, :icend])
            ok &= bool(np.all(x[n][ibl][:, icend:] == 0.0))
    return ok


def adjoint_identity(st, tl, inc, x, tag):
    """<TL dx, TL dx> = <dx, AD TL dx> per column, sums in fp64 over the fp32 fields; PSUPSAT's adjoint is assigned
    PTSPHY*zqp1 by the reference (cloudsc2ad.F90:1733), hence the corrected term."""
    worst = 0.0
    for ibl in range(st.nblocks):
        icend = min(st.nproma, st.ngptot - ibl * st.nproma)
        f = lambda a: a[ibl][:, :icend].astype(np.float64)  # noqa: E731
        n1 = sum((f(tl[n]) ** 2).sum(axis=0) for n in tl)
        n2 = sum((f(inc[n]) * f(x[n])).sum(axis=0) for n in inc if n != "supsat")
        n2 = n2 + (f(inc["supsat"]) * f(x["supsat"]) / st.ptsphy).sum(axis=0)
        worst = max(worst, float(np.max(np.abs(n1 - n2) / np.abs(n1))))
    print(f"  {tag}: adjoint identity, worst column |n1-n2|/n1 = {worst:.3e}  (fp32 eps = 1.19e-07)")
    return worst


def host_checks():
    """csrc/cloudsc2_level.hpp + cloudsc2_column.hpp compiled for the host with -DCLOUDSC2_SINGLE."""
    ok = True
    for flags, precise in ((dict(), 0), (dict(levapls2=True, lregcl=True), 0), (dict(), 1)):
        tab = c2.random_table(137, 40, seed=11)
        prm = make_params(tab, **flags)
        r32, r64 = refs(prm)
        nproma, ngptot = 16, 40
        st = c2.state_from_table(tab, nproma, ngptot)
        assert st.PT.dtype == F32
        nb, nlev = st.nblocks, st.nlev
        qsat = qsat_blocks(r32, st)
        inc = increments_of(st, qsat)
        hc = hostcheck()
        hc.hostcheck_set_precise(precise)
        print(f"host fp32: flags={flags} precise={precise}")
        got = st.copy()
        i, o = host_traj_blocks(got, qsat)
        tl = flat_fields("out", nb, nlev, nproma)
        di, do_ = flat_block("in", inc), flat_block("out", tl)
        assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i), C.byref(o), C.byref(di), C.byref(do_)) == 0
        # y := the fp32 reference's TL outputs, so both adjoints start from identical data
        y = flat_fields("out", nb, nlev, nproma)
        for ibl in range(nb):
            icend = min(nproma, ngptot - ibl * nproma)
            inp = refcall.block_inputs(st, ibl, qsat[ibl])
            dinp = {n: np.ascontiguousarray(inc[n][ibl]) for n in inc}
            for d in (inp, dinp):
                for a in d.values():
                    a[:, icend:] = 1.0
            _, d32 = r32.cloudsc2tl(st.ptsphy, inp, dinp, kfdia=icend, ldrain1d=False)
            for n in y:
                y[n][ibl][:, :icend] = d32[n][:, :icend]
        x = flat_fields("in", nb, nlev, nproma)
        scratch = np.zeros((nb, nlev, nproma), dtype=F32)
        got2 = st.copy()
        i2, o2 = host_traj_blocks(got2, qsat)
        ai, ao = flat_block("in", x), flat_block("out", y)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i2), C.byref(o2), C.byref(ai), C.byref(ao),
                               scratch.ctypes.data) == 0
        ok &= compare_blocks(st, got, tl, x, inc, qsat, r32, r64)
        # identity with OUR TL outputs as y
        y2 = {n: a.copy() for n, a in tl.items()}
        x2 = flat_fields("in", nb, nlev, nproma)
        got3 = st.copy()
        i3, o3 = host_traj_blocks(got3, qsat)
        ai, ao = flat_block("in", x2), flat_block("out", y2)
        assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i3), C.byref(o3), C.byref(ai), C.byref(ao),
                               scratch.ctypes.data) == 0
        ok &= adjoint_identity(st, tl, inc, x2, "host") < 2e-4

        # the 32-bit byte-offset variants of the sweeps (what the GPU launches for buffers < 4 GiB) give the same bits:
        # NL with the driver's zero plane, TL, AD with its checkpoint plane
        hc.hostcheck_set_off32(1)
        try:
            S = nproma * nlev
            nl64, nl32 = st.copy(), st.copy()
            for off32, tgt in ((0, nl64), (1, nl32)):
                hc.hostcheck_set_off32(off32)
                tgt.B_LOC[...] = F32(-5.0)
                ii, oo = host_traj_blocks(tgt, qsat)
                zero = B.Field()
                zero.ptr = tgt.B_LOC.ctypes.data + 4 * 7 * S
                zero.block_stride = 8 * S
                assert hc.hostcheck_nl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(ii), C.byref(oo), zero, 0.0) == 0
                assert np.all(tgt.B_LOC[:, 7] == 0.0) and np.all(tgt.B_LOC[:, [1, 5, 6]] == F32(-5.0))
            for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN"):
                assert np.array_equal(getattr(nl64, n), getattr(nl32, n)), ("off32 NL", n)
            tl32 = flat_fields("out", nb, nlev, nproma)
            got4 = st.copy()
            i4, o4 = host_traj_blocks(got4, qsat)
            di, do_ = flat_block("in", inc), flat_block("out", tl32)
            assert hc.hostcheck_tl(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i4), C.byref(o4), C.byref(di), C.byref(do_)) == 0
            y3 = {n: a.copy() for n, a in tl32.items()}
            x3 = flat_fields("in", nb, nlev, nproma)
            guard = np.full((nb + 1, nlev, nproma), F32(7.0))  # one spare block behind the checkpoint plane
            ai, ao = flat_block("in", x3), flat_block("out", y3)
            assert hc.hostcheck_ad(C.byref(prm), st.ptsphy, nproma, nlev, ngptot, C.byref(i4), C.byref(o4), C.byref(ai), C.byref(ao),
                                   guard.ctypes.data) == 0
            assert np.all(guard[nb] == F32(7.0)), "checkpoint stores ran past the scratch plane"
            for n in tl:
                assert np.array_equal(tl[n], tl32[n]), ("off32 TL", n)
            for n in x2:
                assert np.array_equal(x2[n], x3[n]), ("off32 AD", n)
        finally:
            hc.hostcheck_set_off32(0)
        hc.hostcheck_set_precise(0)
    return ok


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "host"
    if what == "host":
        good = host_checks()
    else:
        from tests.single_checks_gpu import gpu_checks

        good = gpu_checks()
    print("SINGLE CHECKS", "PASSED" if good else "FAILED")
    sys.exit(0 if good else 1)
