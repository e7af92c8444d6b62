"""GPU half of tests/single_checks.py (fp32 kernels of libcloudsc2_hip_sp.so through the C ABI)."""
from __future__ import annotations

import numpy as np

from tests.single_checks import F32, adjoint_identity, compare_blocks, qsat_blocks, refs
from tests.util import B, c2, make_params, refcall, relerr


def _np(ff):
    return {n: t.cpu().numpy() for n, t in ff.t.items()}


def gpu_checks():
    import torch

    assert B.device_available(), "no HIP device"
    ok = True
    for flags, precise in ((dict(), False), (dict(levapls2=True, lregcl=True), False), (dict(), True)):
        B.set_math_mode(precise)
        tab = c2.random_table(137, 300, seed=5)
        prm = make_params(tab, **flags)
        r32, r64 = refs(prm)
        nproma, ngptot = 64, 300  # 5 blocks, ragged tail of 44
        st = c2.state_from_table(tab, nproma, ngptot)
        nb, nlev = st.nblocks, st.nlev
        print(f"gpu fp32: flags={flags} precise={precise}")

        ds = c2.DeviceState(st)
        assert ds.PT.dtype == torch.float32
        ds.satur(prm)
        torch.cuda.synchronize()
        qsat = ds.QSAT.cpu().numpy()
        qref = qsat_blocks(r32, st)
        for ibl in range(nb):
            icend = min(nproma, ngptot - ibl * nproma)
            e = relerr(qref[ibl][:, :icend].astype(np.float64), qsat[ibl][:, :icend].astype(np.float64))
            assert e < 2e-6, ("satur", e)
        # use the reference's PQS so that every later difference is the kernels' own
        ds.QSAT.copy_(torch.from_numpy(qref))
        inc = ds.increments()
        tlo = c2.FlatFields("out", nb, nlev, nproma, ds.device)
        ds.tl(prm, inc, tlo)
        torch.cuda.synchronize()
        got = ds.download(st.copy())
        tl, incn = _np(tlo), _np(inc)

        # the NL kernel (fused SATUR off: same PQS) gives the trajectory the TL kernel stored, to fma-contraction differences
        # between the two separately compiled kernels
        ds2 = c2.DeviceState(st)
        ds2.QSAT.copy_(torch.from_numpy(qref))
        ds2.nl(prm, fused_satur=False)
        torch.cuda.synchronize()
        got_nl = ds2.download(st.copy())
        for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN"):
            a, b = getattr(got, n), getattr(got_nl, n)
            if n == "B_LOC":
                a, b = a[:, [0, 2, 3, 4]], b[:, [0, 2, 3, 4]]
            e = relerr(b.astype(np.float64), a.astype(np.float64))
            if e >= 2e-5:
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                idx = np.unravel_index(np.argmax(d), d.shape)
                print(f"  NL vs TL trajectory {n}: rel {e:.3e} at {idx}: {a[idx]!r} vs {b[idx]!r}; n(|d|>1e-4 max) = "
                      f"{int((d > 1e-4 * np.abs(b).max()).sum())}")
                ok = False

        # AD applied to y = the fp32 reference's TL outputs
        y = c2.FlatFields("out", nb, nlev, nproma, ds.device)
        for ibl in range(nb):
            icend = min(nproma, ngptot - ibl * nproma)
            inp = refcall.block_inputs(st, ibl, qref[ibl])
            dinp = {n: np.ascontiguousarray(incn[n][ibl]) for n in incn}
            for d in (inp, dinp):
                for a in d.values():
                    a[:, icend:] = 1.0
            _, d32 = r32.cloudsc2tl(st.ptsphy, inp, dinp, kfdia=icend)
            for n in d32:
                blk = np.zeros_like(d32[n])
                blk[:, :icend] = d32[n][:, :icend]
                y.t[n][ibl].copy_(torch.from_numpy(blk))
        x = c2.FlatFields("in", nb, nlev, nproma, ds.device)
        ds.ad(prm, x, y, ds.new_scratch())
        torch.cuda.synchronize()
        for n, t in y.t.items():
            assert float(t.abs().max()) == 0.0, ("output adjoint not consumed", n)
        ok &= compare_blocks(st, got, tl, _np(x), incn, qref, r32, r64)

        # adjoint identity with our own TL outputs as y
        y2 = c2.FlatFields("out", nb, nlev, nproma, ds.device)
        for n in y2.t:
            y2.t[n].copy_(tlo.t[n])
        x2 = c2.FlatFields("in", nb, nlev, nproma, ds.device)
        ds.ad(prm, x2, y2, ds.new_scratch())
        torch.cuda.synchronize()
        ok &= adjoint_identity(st, tl, incn, _np(x2), "gpu") < 2e-4
    B.set_math_mode(False)

    # driver level (host arrays, slab-pipelined transfers) = kernel level, bit for bit; untouched planes keep their values
    tab = c2.synthetic_table()
    prm = make_params(tab)
    st = c2.state_from_table(tab, 128, 1000, poison_outputs=-5.0)
    ref = st.copy()
    ms = c2.run_state(prm, st, "nl")
    ds = c2.DeviceState(ref)
    ds.nl(prm)
    torch.cuda.synchronize()
    ds.download(ref)
    for n in ("PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN"):
        assert np.array_equal(getattr(st, n), getattr(ref, n)), n
    assert np.array_equal(st.B_LOC[:, [0, 2, 3, 4, 7]], ref.B_LOC[:, [0, 2, 3, 4, 7]])
    assert np.all(st.B_LOC[:, [1, 5, 6]] == F32(-5.0))
    print(f"driver-level NL == kernel-level NL (kernel {ms:.3f} ms for 1000 columns)")

    # The adjoint test driver in fp32.  norm3 is in units of the FP64 epsilon whatever JPRB is ("machine precision is defined
    # here as strictly 64bits", cloudsc_driver_ad_mod.F90:258-264), so an fp32 build cannot pass the reference's verdict
    # (< 10000, :289): the reference's own -DSINGLE binary reports ~3e9 and prints TEST FAILED, and so does this library.
    # What the fp32 build IS held to is the same threshold in units of its own epsilon -- a separate, fp32-only criterion.
    st = c2.state_from_table(tab, 100, 100)
    zn, passed, _ = c2.run_state(prm, st, "ad")
    zn32 = zn * (2.220446049250313e-16 / 1.1920928955078125e-07)
    print(f"CLOUDSC_DRIVER_AD fp32: max norm3 = {zn:.3e} x EPSILON(1._8) -> reference verdict "
          f"{'TEST OK' if passed else 'TEST FAILED'} (the reference's -DSINGLE binary: TEST FAILED)")
    print(f"fp32 criterion: max norm3 = {zn32:.3f} x EPSILON(1._4) -> {'fp32 criterion OK' if zn32 < 1e4 else 'fp32 criterion FAILED'}")
    ok &= (not passed) and zn32 < 1e4
    try:
        znormg, tpass, itest, _ = c2.run_state(prm, c2.state_from_table(tab, 100, 100), "tl")
        print("CLOUDSC_DRIVER_TL fp32 (informative: the V shape needs fp64 head-room):", np.array2string(znormg, precision=3),
              "PASSED" if tpass else f"FAILED itest={itest}")
    except c2.Cloudsc2Error as e:
        print("CLOUDSC_DRIVER_TL fp32 (informative):", e)

    # The TL sweep that forms the test drivers' increments itself (cloudsc2_tl_launch_self) and the Taylor test's lambda sweep
    # (cloudsc2_taylor_sweep_launch: ten lambdas on the lanes of a wave) in fp32, against the forms they replace
    tabs = c2.synthetic_table()
    prms = make_params(tabs, lregcl=False)
    ds = c2.DeviceState.from_table(tabs, 64, 500)
    ds.satur(prms)
    ds.nl(prms, fused_satur=False)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    _, dself = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.increments(into=inc)
    ds.tl(prms, inc, dout, store_traj=False)
    ds.tl(prms, None, dself, store_traj=False)
    sweep = ds.taylor_sweep(prms, dout)
    torch.cuda.synchronize()
    for n in B.OUT_NAMES:
        a, b = dout.t[n].double(), dself.t[n].double()
        e = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-300)
        floor_ok = float((a - b).abs().max()) < 1e-9 and n in ("clc", "covptot")
        assert e < 2e-5 or floor_ok, ("tl self-increment fp32", n, e)
    from dwarf_p_cloudsc2_tl_ad_amd.driver import _fld
    from dwarf_p_cloudsc2_tl_ad_amd.state import PLANE_Q, PLANE_QI, PLANE_QL, PLANE_T

    pc = {n: torch.zeros_like(getattr(ds, n)) for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")}
    S, H = ds.nproma * ds.nlev, ds.nproma * (ds.nlev + 1)
    po = B.Outputs()
    po.tent, po.tenq = _fld(pc["B_LOC"], PLANE_T * S, 8 * S), _fld(pc["B_LOC"], PLANE_Q * S, 8 * S)
    po.tenl, po.teni = _fld(pc["B_LOC"], PLANE_QL * S, 8 * S), _fld(pc["B_LOC"], PLANE_QI * S, 8 * S)
    po.clc, po.covptot = _fld(pc["PA"], 0, S), _fld(pc["PCOVPTOT"], 0, S)
    po.fplsl, po.fplsn = _fld(pc["PFPLSL"], 0, H), _fld(pc["PFPLSN"], 0, H)
    po.fhpsl, po.fhpsn = _fld(pc["PFHPSL"], 0, H), _fld(pc["PFHPSN"], 0, H)
    size = [float(t.abs().max()) for t in (ds.B_LOC[:, PLANE_T], ds.B_LOC[:, PLANE_Q], ds.B_LOC[:, PLANE_QL], ds.B_LOC[:, PLANE_QI], ds.PA,
                                           ds.PFPLSL, ds.PFPLSN, ds.PFHPSL, ds.PFHPSN, ds.PCOVPTOT)]
    worst = 0.0
    for il in range(10):
        lam = 10.0 ** -(il + 1)
        ds.nl(prms, fused_satur=False, pert_lambda=lam, outputs=po)
        old = ds.taylor_sums(po, dout, lam)
        torch.cuda.synchronize()
        got, ref = sweep[il].cpu().numpy(), old.cpu().numpy()
        assert np.all(np.abs(got[:, :, 1] - ref[:, :, 1]) <= 1e-5 * np.abs(ref[:, :, 1]) + 1e-6 * np.abs(ref[:, :, 1]).max(axis=0, keepdims=True) + 1e-30), ("lambda sweep fp32: TL sums", il)
        for f in range(10):
            bound = 64 * 1.1920928955078125e-07 * size[f] * ds.nlev * ds.nproma + 1e-30
            d = float(np.abs(got[:, f, 0] - ref[:, f, 0]).max())
            worst = max(worst, d / bound)
            assert d <= bound, ("lambda sweep fp32", il, f, d, bound)
    print(f"fp32 TL with increments formed in the sweep == TL fed with them; lambda sweep == ten perturbed runs (worst {worst:.3f} of the rounding bound)")

    # device-side tiling and validation in fp32
    tab = c2.random_table(137, 100, seed=3)
    prm = make_params(tab)
    a = c2.DeviceState.from_table(tab, 96, 1000)
    b = c2.DeviceState(c2.state_from_table(tab, 96, 1000))
    for n in a.FULL + a.HALF + ("B_CML", "PCLV"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    a.nl(prm)
    torch.cuda.synchronize()
    from dwarf_p_cloudsc2_tl_ad_amd import fileio

    host = a.download(c2.state_from_table(tab, 96, 1000))
    reft = fileio.reference_table_from_state(host, 100)
    rows, text = a.validate(reft)
    for name, _, s in rows:
        assert s[2] == 0.0 and s[3] == 0.0, (name, s)
    print(text.splitlines()[0])
    print(text.splitlines()[1])

    # the fp32 adjoint takes the two-kernel form up to 400 000 columns and the fused kernel above: same columns, same answers
    tab = c2.synthetic_table()
    prm = make_params(tab, lregcl=True)
    res = []
    for ncols in (1024, 409600):
        d = c2.DeviceState.from_table(tab, 128, ncols)
        d.satur(prm)
        dx = d.increments(zero_supsat=True)
        dy = c2.FlatFields("out", d.nb, d.nlev, d.nproma, d.device)
        d.tl(prm, dx, dy)
        ax = c2.FlatFields("in", d.nb, d.nlev, d.nproma, d.device)
        d.ad(prm, ax, dy, d.new_scratch())
        torch.cuda.synchronize()
        res.append({n: t[:8].cpu().numpy() for n, t in ax.t.items()})  # the first 1024 columns
        del d, dx, dy, ax
    for n in res[0]:
        e = relerr(res[0][n].astype(np.float64), res[1][n].astype(np.float64))
        assert e < 2e-5, ("fused vs two-kernel adjoint", n, e)
    print("fp32 adjoint: fused kernel (409600 columns) == two-kernel form (1024 columns) on the same columns")

    # throughput (informative)
    n = 160000
    tab = c2.synthetic_table()
    prm = make_params(tab)
    ds = c2.DeviceState.from_table(tab, 128, n)
    for _ in range(3):
        ds.nl(prm)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        ds.nl(prm)
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 10
    gbs = c2.bytes_per_column(137) * n / t / 1e6
    print(f"fp32 NL {n} columns: {t:.3f} ms, {n / t * 1e3:.3e} columns/s, {gbs:.0f} GB/s algorithmic")
    return ok
