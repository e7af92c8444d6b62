"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU checker on the same seeded inputs.

Checker = the unmodified reference Fortran (oracle/_ref/libcloudsc2_ref.so) when it travelled to the box, otherwise
the C restatement (oracle/libcloudsc2_oracle.so).  Tolerances: BASELINE.json's north_star asks for NL within 1e-10
relative in fp64; the kernels run with FMA contraction and the device libm, the checker without FMA on the host
libm, so agreement is expected at ~1e-13.  Any difference much larger than that is a flipped branch or a logic error,
not rounding, so the tests assert 1e-12 for NL (measured <= 5e-14) and 1e-11 for TL and AD (sums of many
cancelling terms; measured <= 4e-12 over the fuzz cases of tests/fuzz_parity.py) -- two orders tighter than the 1e-10
BASELINE.json asks for.
"""
from __future__ import annotations

import os

import numpy as np
import pytest

from tests.util import B, c2, flat_fields, refcall, relerr, set_lib_params

pytestmark = pytest.mark.gpu

NL_TOL = 1e-12
TLAD_TOL = 1e-11


def checker():
    if refcall.have_ref():
        return refcall.RefLib()
    return refcall.OracleLib()


def ref_qsat(chk, st):
    qsat = np.zeros_like(st.PAP)
    for ibl in range(st.nblocks):
        icend = min(st.nproma, st.ngptot - ibl * st.nproma)
        qsat[ibl] = chk.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        qsat[ibl][:, icend:] = 0.0
    return qsat


def ref_nl_state(chk, st, prm, qsat=None):
    """Reference outputs for a whole state: per-block SATUR + CLOUDSC2 exactly like cloudsc_driver_mod.F90:82-111."""
    out = st.copy()
    out.PCOVPTOT[...] = 0.0
    out.B_LOC[:, 7] = 0.0
    for ibl in range(st.nblocks):
        icend = min(st.nproma, st.ngptot - ibl * st.nproma)
        qs = qsat[ibl] if qsat is not None else chk.satur(np.ascontiguousarray(st.PAP[ibl]), np.ascontiguousarray(st.PT[ibl]), kfdia=icend)
        o = chk.cloudsc2(st.ptsphy, refcall.block_inputs(st, ibl, qs), kfdia=icend, ldrain1d=bool(prm.ldrain1d))
        for n, a in refcall.state_outputs_block(out, ibl).items():
            a[:, :icend] = o[n][:, :icend]
    return out


def assert_outputs_close(ref_st, got_st, tol):
    for n, r in ref_st.outputs().items():
        g = got_st.outputs()[n]
        assert np.all(np.isfinite(g)), n
        assert relerr(r, g) <= tol, (n, relerr(r, g))
        assert c2.validate_l1(r, g) <= tol, (n, c2.validate_l1(r, g))


@pytest.mark.parametrize("nproma,ngptot", [(32, 100), (64, 100), (100, 100), (128, 300), (1, 7), (256, 1000), (1000, 2500),
                                           (512, 100), (64, 1), (128, 129)])
def test_nl_driver_matches_checker(nproma, ngptot):
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, nproma, ngptot, poison_outputs=777.0)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    assert_outputs_close(want, got, NL_TOL)
    # what the kernels must NOT touch: padded tail columns (except PCOVPTOT / CLD(:,:,NCLV) zeroing), B_LOC planes A, QR, QS
    for plane in (1, 5, 6):
        assert np.all(got.B_LOC[:, plane] == 777.0)
    assert np.all(got.B_LOC[:, 7] == 0.0) and np.all(got.PCOVPTOT == 0.0)
    tail = st.nblocks * nproma - ngptot
    if tail:
        assert np.all(got.PFPLSN[-1][:, nproma - tail:] == 777.0)
        assert np.all(got.B_LOC[-1, 0][:, nproma - tail:] == 777.0)


@pytest.mark.parametrize("flags", [dict(levapls2=True), dict(ldrain1d=True)])
def test_nl_evaporation_branch(flags):
    """LEVAPLS2 / LDRAIN1D switch on the precipitation-evaporation block that is dead in the shipped configs
    (cloudsc2.F90:556-591); PCOVPTOT then becomes non-trivial."""
    tab = c2.random_table(137, 64, seed=11)
    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, 64, 128)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    assert np.any(want.PCOVPTOT != 0.0)
    assert_outputs_close(want, got, NL_TOL)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_nl_random_atmospheres(seed):
    tab = c2.random_table(137, 100, seed=seed)
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, 50, 200)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    assert_outputs_close(want, got, NL_TOL)


def _device_tl_ad(tab, nproma, ngptot, flags):
    """Run TL then AD on the GPU at kernel level (device pointers) and with the checker; returns both result sets."""
    import torch

    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, nproma, ngptot)
    nb, nlev = st.nblocks, st.nlev
    chk = checker()
    set_lib_params(chk, prm)
    qsat = ref_qsat(chk, st)

    ds = c2.DeviceState(st, "cuda:0")
    ds.satur(prm)
    torch.cuda.synchronize()
    q_dev = ds.QSAT.cpu().numpy()
    for ibl in range(nb):
        icend = min(nproma, ngptot - ibl * nproma)
        assert relerr(qsat[ibl][:, :icend], q_dev[ibl][:, :icend]) < 1e-14
    inc = ds.increments()
    tl_out = c2.FlatFields("out", nb, nlev, nproma, ds.device)
    ds.tl(prm, inc, tl_out)
    torch.cuda.synchronize()
    tl_dev = {n: t.cpu().numpy() for n, t in tl_out.t.items()}
    traj_dev = ds.download(st.copy())

    # checker TL
    inc_h = {n: t.cpu().numpy() for n, t in inc.t.items()}
    tl_ref = flat_fields("out", nb, nlev, nproma)
    traj_ref = st.copy()
    for ibl in range(nb):
        icend = min(nproma, ngptot - ibl * nproma)
        dinp = {n: np.ascontiguousarray(inc_h[n][ibl]) for n in inc_h}
        inp = refcall.block_inputs(st, ibl, qsat[ibl])
        for d in (inp, dinp):  # keep the checker away from the uninitialised tail
            for a in d.values():
                a[:, icend:] = 1.0
        o5, do = chk.cloudsc2tl(st.ptsphy, inp, dinp, kfdia=icend, ldrain1d=bool(prm.ldrain1d))
        for n in do:
            tl_ref[n][ibl][:, :icend] = do[n][:, :icend]
        for n, a in refcall.state_outputs_block(traj_ref, ibl).items():
            a[:, :icend] = o5[n][:, :icend]

    # AD with y = TL output (the adjoint test's choice, cloudsc_driver_ad_mod.F90:216-237), x pre-filled with a
    # non-zero background to check accumulation
    rng = np.random.default_rng(5)
    x0 = {n: rng.standard_normal(a.shape) * (np.abs(a).max() + 1e-30) for n, a in inc_h.items()}
    adj_in = c2.FlatFields("in", nb, nlev, nproma, ds.device)
    for n in x0:
        adj_in.t[n].copy_(torch.from_numpy(x0[n]))
    scratch = ds.new_scratch()
    ds.ad(prm, adj_in, tl_out, scratch)
    torch.cuda.synchronize()
    x_dev = {n: t.cpu().numpy() for n, t in adj_in.t.items()}
    y_after = {n: t.cpu().numpy() for n, t in tl_out.t.items()}

    x_ref = {n: a.copy() for n, a in x0.items()}
    for ibl in range(nb):
        icend = min(nproma, ngptot - ibl * nproma)
        inp = refcall.block_inputs(st, ibl, qsat[ibl])
        for a in inp.values():
            a[:, icend:] = 1.0
        ain = {n: x_ref[n][ibl].copy() for n in x_ref}
        aout = {n: tl_ref[n][ibl].copy() for n in tl_ref}  # consumed (zeroed) by the AD
        chk.cloudsc2ad(st.ptsphy, inp, ain, aout, kfdia=icend, ldrain1d=bool(prm.ldrain1d))
        for n in ain:
            x_ref[n][ibl][:, :icend] = ain[n][:, :icend]
    return dict(st=st, tl_dev=tl_dev, tl_ref=tl_ref, traj_dev=traj_dev, traj_ref=traj_ref, x_dev=x_dev, x_ref=x_ref,
                x0=x0, y_after=y_after, inc=inc_h)


@pytest.mark.parametrize("flags", [dict(), dict(lregcl=True), dict(levapls2=True), dict(levapls2=True, lregcl=True)])
@pytest.mark.parametrize("nproma,ngptot", [(64, 100), (100, 100)])
def test_tl_ad_kernels_match_checker(flags, nproma, ngptot):
    tab = c2.random_table(137, 100, seed=7) if flags.get("levapls2") else c2.synthetic_table()
    r = _device_tl_ad(tab, nproma, ngptot, flags)
    st = r["st"]
    act = lambda a: np.concatenate([a[ibl][:, : min(nproma, ngptot - ibl * nproma)] for ibl in range(st.nblocks)], axis=1)  # noqa: E731
    for n in r["tl_ref"]:
        assert relerr(act(r["tl_ref"][n]), act(r["tl_dev"][n])) <= TLAD_TOL, ("tl", n)
    assert_outputs_close(r["traj_ref"], r["traj_dev"], NL_TOL)
    for n in r["x_ref"]:
        ref_inc = act(r["x_ref"][n]) - act(r["x0"][n]) if n != "supsat" else act(r["x_ref"][n])
        got_inc = act(r["x_dev"][n]) - act(r["x0"][n]) if n != "supsat" else act(r["x_dev"][n])
        scale = max(np.abs(act(r["x_ref"][n])).max(), 1e-300)
        assert np.abs(got_inc - ref_inc).max() / scale <= TLAD_TOL, ("ad", n)
    # output adjoints are consumed (zeroed) for active columns
    for n, a in r["y_after"].items():
        assert np.all(act(a) == 0.0), n
    # the identity the adjoint test is built on: <TL x, TL x> = <x, AD(TL x)> per column, with a zero background
    # (PSUPSAT excluded: the reference assigns its adjoint with a spurious PTSPHY factor, cloudsc2ad.F90:1733)


@pytest.mark.parametrize("flags", [dict(lregcl=True), dict(levapls2=True, lregcl=True)])
@pytest.mark.parametrize("assign", [False, True])
def test_adjoint_as_two_sweeps_equals_the_whole(flags, assign):
    """cloudsc2_ad_launch_forward + cloudsc2_ad_launch_reverse = cloudsc2_ad_launch bit for bit (CLOUDSC2AD is its forward sweep,
    cloudsc2ad.F90:366-866, followed by its reverse sweep, :877-1740); the reverse sweep ALONE after a TL launch that stored the
    trajectory outputs gives the same again (what cloudsc2_ad_symmetry_run does); and without the evaporation branch the
    cover-checkpoint plane is not needed at all (NULL) -- with it, a missing plane is refused."""
    import torch

    tab = c2.random_table(137, 100, seed=11) if flags.get("levapls2") else c2.synthetic_table()
    nproma, ngptot = 64, 300
    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, nproma, ngptot)
    evap = bool(flags.get("levapls2"))

    def run(mode):
        ds = c2.DeviceState(st, "cuda:0")
        ds.satur(prm)
        inc = ds.increments(zero_supsat=True)
        y = c2.FlatFields("out", ds.nb, ds.nlev, nproma, ds.device)
        ds.tl(prm, inc, y)  # trajectory outputs stored: PFPLSL5 / PFPLSN5 are in the state afterwards
        x = c2.FlatFields("in", ds.nb, ds.nlev, nproma, ds.device)
        for t in x.t.values():
            t.fill_(0.5)
        scratch = ds.new_scratch() if (evap or mode == "whole") else None
        if mode == "whole":
            ds.ad(prm, x, y, scratch, assign=assign)
        elif mode == "two":
            ds.ad(prm, x, y, scratch, sweep="forward")
            ds.ad(prm, x, y, scratch, assign=assign, sweep="reverse")
        else:  # reverse sweep alone on what the TL launch left (with the evaporation branch the checkpoints must exist)
            if evap:
                ds.ad(prm, x, y, scratch, sweep="forward")
                for n in ("B_LOC", "PA", "PCOVPTOT", "PFHPSL", "PFHPSN"):
                    getattr(ds, n).fill_(float("nan"))  # nothing of the trajectory outputs but the two flux carries is read
            ds.ad(prm, x, y, scratch, assign=assign, sweep="reverse")
        torch.cuda.synchronize()
        return {n: t.cpu().numpy() for n, t in x.t.items()}, {n: t.cpu().numpy() for n, t in y.t.items()}

    xw, yw = run("whole")
    for mode in ("two", "reverse"):
        xm, ym = run(mode)
        for n in xw:
            assert np.array_equal(xw[n], xm[n]), (mode, n)
        for n in yw:
            assert np.array_equal(yw[n], ym[n]), (mode, n)
    if evap:
        ds = c2.DeviceState(st, "cuda:0")
        x = c2.FlatFields("in", ds.nb, ds.nlev, nproma, ds.device)
        y = c2.FlatFields("out", ds.nb, ds.nlev, nproma, ds.device)
        with pytest.raises(RuntimeError, match="checkpoint"):
            ds.ad(prm, x, y, None, sweep="reverse")


@pytest.mark.parametrize("nlev", [11, 60, 200])
def test_other_numbers_of_levels(nlev):
    """NLEV other than 137 (the kernels take it at run time; CETA holds up to 200 levels): NL through the driver, TL and
    AD at kernel level, against the checker.  With 11 levels the tropopause band is empty."""
    tab = c2.random_table(nlev, 64, seed=9)
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, 64, 100, poison_outputs=3.0)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    assert_outputs_close(want, got, NL_TOL)
    r = _device_tl_ad(tab, 64, 100, {})
    act = lambda a: np.concatenate([a[ibl][:, : min(64, 100 - ibl * 64)] for ibl in range(2)], axis=1)  # noqa: E731
    for n in r["tl_ref"]:
        assert relerr(act(r["tl_ref"][n]), act(r["tl_dev"][n])) <= TLAD_TOL, ("tl", n)
    for n in r["x_ref"]:
        scale = max(np.abs(act(r["x_ref"][n])).max(), 1e-300)
        assert np.abs(act(r["x_dev"][n]) - act(r["x_ref"][n])).max() / scale <= TLAD_TOL, ("ad", n)


def test_taylor_test_passes_on_gpu():
    """CLOUDSC_DRIVER_TL semantics (cloudsc_driver_tl_mod.F90:272-311): V-shaped convergence of the Taylor ratio."""
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    for nproma, ngptot in ((32, 100), (1, 100)):
        st = c2.state_from_table(tab, nproma, ngptot)
        zn, ok, itest, _ = c2.run_state(prm, st, "tl")
        assert np.all(np.isfinite(zn))
        assert ok, (nproma, zn, itest)
        assert np.min(np.abs(1.0 - zn)) < 1e-5


def test_taylor_ratios_match_the_reference_driver():
    """tests/golden/drivers.json holds what the reference's own CLOUDSC_DRIVER_TL printed for the same inputs.  The
    ratios are cancellation-dominated below lambda = 1e-7, so only the first six are compared tightly."""
    import json
    import os

    from tests.util import ROOT

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "drivers.json")))["drivers"]
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    for nproma in (32, 1):
        st = c2.state_from_table(tab, nproma, 100)
        zn, ok, itest, _ = c2.run_state(prm, st, "tl")
        ref = gold[f"tl_nproma{nproma}_ngptot100"]
        assert np.allclose(zn[:6], ref["znormg"][:6], rtol=1e-6, atol=0), (nproma, zn, ref["znormg"])
        assert ok and ref["verdict"].endswith(str(itest)), (nproma, itest, ref["verdict"])


def _golden_table(which):
    """"library": cloudsc2_synthetic_table (what every front end loads); "numpy": the same recipe as numpy evaluated it in rounds 1-4
    (tests/golden/table_numpy.npz holds the fields that differ, in the last place)."""
    from tests.util import ROOT

    tab = c2.synthetic_table()
    if which == "numpy":
        z = np.load(os.path.join(ROOT, "tests", "golden", "table_numpy.npz"))
        for n in z.files:
            assert tab[n].shape == z[n].shape and not np.array_equal(tab[n], z[n]), n
            tab[n] = z[n]
    return tab


def _taylor_lines(out):
    import re

    zn = [float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", out, flags=re.M)][:10]
    return np.array(zn), re.search(r"TEST (PASSED|FAILLED).*", out).group(0).strip()


@pytest.mark.parametrize("which, nproma, ngptot", [("library", 32, 800), ("library", 128, 3200), ("numpy", 32, 800), ("numpy", 128, 3200)])
def test_taylor_verdict_over_the_full_block_set(which, nproma, ngptot, tmp_path):
    """The Taylor statistic is a MAX over NPROMA blocks (cloudsc_driver_tl_mod.F90:21-31,249), and the 100-periodic state has
    lcm(NPROMA, 100) / NPROMA distinct blocks: 800 columns at NPROMA 32, 3200 at 128 -- the statistic of any larger run, 160 000
    columns included.  tests/golden/drivers.json holds what the reference's own CLOUDSC_DRIVER_TL prints there, on two tables that
    differ in the last place only: on the library's the test PASSES at both blockings, on the numpy-flavoured one it FAILS at NPROMA 32
    (err 10: one wiggle in the round-off arm of the V-shape test; round 4's logs).  Every front end must reproduce
    the first six ratios to 1e-6 and the SAME verdict, pass or fail: the host-array driver (run_state = cloudsc2_tl_taylor_run), the
    resident state (cloudsc2_state_tl_taylor), and the Fortran main dwarf-cloudsc2-tl in both of its modes (the table reaches it
    through input.h5)."""
    import json

    from dwarf_p_cloudsc2_tl_ad_amd import fileio
    from tests.util import ROOT

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "drivers.json")))["drivers"]
    ref = gold[f"tl_nproma{nproma}_ngptot{ngptot}" + ("_numpy_table" if which == "numpy" else "")]
    want, want_ok = np.array(ref["znormg"]), ref["verdict"].startswith("TEST PASSED")
    assert want_ok == (not (which == "numpy" and nproma == 32)), ref  # the fixture itself: one failing verdict on record
    tab = _golden_table(which)
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    results = {}
    zn, ok, itest, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, ngptot), "tl")
    results["cloudsc2_tl_taylor_run"] = (zn, ok, itest)
    zn, ok, itest, _ = c2.ResidentState.from_table(tab, nproma, ngptot).tl_taylor(prm)
    results["cloudsc2_state_tl_taylor"] = (zn, ok, itest)
    prm.nlev = 137
    fileio.write_input_file(str(tmp_path / "input.h5"), tab, prm)
    for mode, env in (("host arrays", {"CLOUDSC2_RESIDENT": "0"}), ("resident", {"CLOUDSC2_RESIDENT": "1"})):
        out, _ = _run_fortran("dwarf-cloudsc2-tl", 1, ngptot, nproma, cwd=str(tmp_path), env={**env, "CLOUDSC2_MATH": ""})
        zn, verdict = _taylor_lines(out)
        results[f"dwarf-cloudsc2-tl, {mode}"] = (zn, verdict.startswith("TEST PASSED"), int(verdict.split()[-1]))
    print(which, nproma, ngptot, "reference:", ref["verdict"], {k: (v[1], v[2]) for k, v in results.items()})
    for name, (zn, ok, itest) in results.items():
        assert zn.shape == (10,) and np.allclose(zn[:6], want[:6], rtol=1e-6, atol=0), (name, zn, want)
        assert bool(ok) == want_ok and ref["verdict"].endswith(str(itest)), (name, ok, itest, ref["verdict"], zn, want)
    # the same library on the same inputs: the Fortran main prints what the Python harness computes, digit for digit
    for a, b in (("cloudsc2_tl_taylor_run", "dwarf-cloudsc2-tl, host arrays"), ("cloudsc2_state_tl_taylor", "dwarf-cloudsc2-tl, resident")):
        assert np.allclose(results[a][0], results[b][0], rtol=1e-14, atol=0), (a, results[a][0], results[b][0])


def test_taylor_verdict_of_main_and_harness_agree_at_160000_columns():
    """Round 4's open question (profiles/EXPERIMENTS.md section 8): at 160 000 columns x NPROMA 32 the Python harness failed the Taylor
    test where the Fortran main on the same library passed.  Neither evaluated another statistic: their synthetic TABLES differed in
    the last place (numpy's exp / power against flang's = glibc's), and the verdict is decided by round-off.  With the table in the
    library (cloudsc2_synthetic_table) both load the same bits: same ratios, same verdict -- the reference's own for the full block
    set at that blocking (tests/golden/drivers.json, NPROMA 32 x 800 columns)."""
    import json

    from tests.util import ROOT

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "drivers.json")))["drivers"]["tl_nproma32_ngptot800"]
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    zn, ok, itest, _ = c2.ResidentState.from_table(tab, 32, 160000).tl_taylor(prm)
    out, _ = _run_fortran("dwarf-cloudsc2-tl", 1, 160000, 32, env={"CLOUDSC2_RESIDENT": "1", "CLOUDSC2_MATH": ""})  # (no input.h5: the built-in table)
    zf, verdict = _taylor_lines(out)
    print("160000 x 32: harness", zn, ok, itest, "| main", verdict, "| reference at 800 x 32:", gold["verdict"])
    assert np.allclose(zn, zf, rtol=1e-14, atol=0), (zn, zf)
    assert bool(ok) == verdict.startswith("TEST PASSED") and verdict.endswith(str(itest))
    assert np.allclose(zn[:6], gold["znormg"][:6], rtol=1e-6, atol=0)
    assert bool(ok) == gold["verdict"].startswith("TEST PASSED") and gold["verdict"].endswith(str(itest)), (ok, itest, gold["verdict"])


def test_adjoint_symmetry_on_gpu():
    """CLOUDSC_DRIVER_AD semantics (cloudsc_driver_ad_mod.F90:286-294): max_col |<TLx,TLx> - <x,AD TLx>| / (eps <x,AD TLx>) < 1e4."""
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    for nproma, ngptot in ((100, 100), (64, 16384)):
        st = c2.state_from_table(tab, nproma, ngptot)
        zad, ok, _ = c2.run_state(prm, st, "ad")
        assert np.isfinite(zad) and ok, (nproma, ngptot, zad)
        assert zad * np.finfo(np.float64).eps < 1e-12, zad  # BASELINE configs[3]: <TLx,y> = <x,ADy> to 1e-12


def test_full_size_periodicity_and_determinism():
    """BASELINE size (NGPTOT=160000): the inputs are a periodic tiling of 100 columns (expand_mod.F90:283-296), so the
    outputs of column g must equal those of column g mod 100 bit for bit, for every NPROMA, and two runs must agree."""
    import torch

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    base = None
    for nproma in (32, 128):
        ngptot = 160000
        st = c2.state_from_table(tab, nproma, ngptot)
        ds = c2.DeviceState(st, "cuda:0")
        ds.nl(prm)
        torch.cuda.synchronize()
        first = ds.PFPLSN.clone()
        ds.nl(prm)
        torch.cuda.synchronize()
        assert torch.equal(first, ds.PFPLSN)
        got = ds.download(st)
        for name, a in got.outputs().items():
            nlevx = a.shape[1]
            cols = a.transpose(0, 2, 1).reshape(-1, nlevx)[:ngptot]  # (column, level)
            ref = cols[:100]
            assert np.array_equal(cols.reshape(-1, 100, nlevx), np.broadcast_to(ref, (ngptot // 100, 100, nlevx))), name
            if base is None:
                continue
            assert np.array_equal(ref, base[name]), (name, nproma)
        if base is None:
            base = {n: a.transpose(0, 2, 1).reshape(-1, a.shape[1])[:100].copy() for n, a in got.outputs().items()}


def _periodic(c, ngptot):
    """column g == column g mod 100, bit for bit, for ALL columns of a (column, level) tensor (any NGPTOT)"""
    import torch

    if ngptot % 100 == 0:
        return torch.equal(c.reshape(-1, 100, c.shape[1]), c[:100].expand(ngptot // 100, 100, c.shape[1]))
    idx = torch.arange(ngptot, device=c.device) % 100
    return torch.equal(c, c[:100][idx])


def _full_size_tl_ad(ngptot, linearity):
    """The TL and the AD kernels at a full size through size-independent properties -- periodicity (column g == column g mod 100,
    bit for bit), linearity of the TL in dx, the adjoint identity <TL dx, TL dx> = <dx, AD(TL dx)> per column (the reference's own
    test, cloudsc_driver_ad_mod.F90:184-264) -- AND directly against CLOUDSC2TL / CLOUDSC2AD for a sample of blocks."""
    import torch

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    nproma = 128
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.satur(prm)
    dx = ds.increments(zero_supsat=True)
    dy = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.tl(prm, dx, dy)
    torch.cuda.synchronize()

    def columns(t):  # (NBLOCKS, NLEVx, NPROMA) -> (column, level)
        return t.permute(0, 2, 1).reshape(-1, t.shape[1])[:ngptot]

    # periodicity of the TL outputs
    for n, t in dy.t.items():
        assert _periodic(columns(t), ngptot), n
    if linearity:  # TL(2 dx) == 2 TL(dx) (scaling by a power of two is exact in every product and sum)
        dx2 = c2.FlatFields("in", ds.nb, ds.nlev, ds.nproma, ds.device)
        for n in dx.t:
            torch.mul(dx.t[n], 2.0, out=dx2.t[n])
        dy2 = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
        ds.tl(prm, dx2, dy2)
        torch.cuda.synchronize()
        for n in dy.t:
            assert torch.equal(dy2.t[n], 2.0 * dy.t[n]), n
        del dx2, dy2
    # adjoint identity per column; the AD consumes (zeroes) its output adjoints and accumulates into zeroed input adjoints
    norm1 = sum((columns(t) ** 2).sum(dim=1) for t in dy.t.values())
    sample = (0, 1, ds.nb // 2, ds.nb - 1)  # blocks compared DIRECTLY with the checker below (block 0 alone holds all 100 distinct columns)
    tl_dev = {ibl: {n: t[ibl].cpu().numpy() for n, t in dy.t.items()} for ibl in sample}
    xa = c2.FlatFields("in", ds.nb, ds.nlev, ds.nproma, ds.device)
    scratch = ds.new_scratch()
    ds.ad(prm, xa, dy, scratch)
    torch.cuda.synchronize()
    for n, t in dy.t.items():
        assert not t.any(), n
    norm2 = sum((columns(dx.t[n]) * columns(xa.t[n])).sum(dim=1) for n in dx.t if n != "supsat")
    err = ((norm1 - norm2).abs() / norm2.abs()).max().item() / np.finfo(np.float64).eps
    assert err < 1e4, err          # the reference's threshold (cloudsc_driver_ad_mod.F90:289)
    assert err * np.finfo(np.float64).eps < 1e-12, err  # BASELINE configs[3]: to 1e-12
    for n, t in xa.t.items():
        assert _periodic(columns(t), ngptot), n

    # The same launches against CLOUDSC2TL / CLOUDSC2AD themselves (cloudsc2tl.F90:10-24, cloudsc2ad.F90:10-24), block by block
    # for a sample of blocks: every TL output and every input adjoint of every column of those blocks.  Together with the
    # bit-periodicity asserted above (column g == column g mod 100 for ALL columns of every TL output and every input adjoint, and
    # block 0 holds columns 0..99) this is the oracle comparison of the exact size, not a property.
    chk = checker()
    set_lib_params(chk, prm)
    worst_tl = worst_ad = 0.0
    for ibl in sample:
        ncol = min(nproma, ngptot - ibl * nproma)
        stb = c2.state_from_table(tab, nproma, nproma, col0=ibl * nproma)  # the host copy of this block's inputs
        qs = ref_qsat(chk, stb)[0]
        dinp = {n: np.ascontiguousarray(t[ibl].cpu().numpy()) for n, t in dx.t.items()}
        o5, do = chk.cloudsc2tl(stb.ptsphy, refcall.block_inputs(stb, 0, qs), dinp, kfdia=ncol, ldrain1d=False)
        for n in do:
            worst_tl = max(worst_tl, relerr(do[n][:, :ncol], tl_dev[ibl][n][:, :ncol]))
        ain = {n: np.zeros_like(a) for n, a in dinp.items()}
        aout = {n: a.copy() for n, a in do.items()}
        chk.cloudsc2ad(stb.ptsphy, refcall.block_inputs(stb, 0, qs), ain, aout, kfdia=ncol, ldrain1d=False)
        for n in ain:
            got = xa.t[n][ibl].cpu().numpy()
            worst_ad = max(worst_ad, np.abs(got - ain[n])[:, :ncol].max() / max(np.abs(ain[n]).max(), 1e-300))
    print(f"{ngptot} columns, blocks {sample} against CLOUDSC2TL / CLOUDSC2AD: worst TL {worst_tl:.1e}, worst AD {worst_ad:.1e}; adjoint identity {err:.1f} eps")
    assert worst_tl <= TLAD_TOL and worst_ad <= TLAD_TOL, (worst_tl, worst_ad)


def test_full_size_tl_ad_properties():
    """BASELINE size (NGPTOT = 160 000)."""
    _full_size_tl_ad(160000, linearity=True)


def test_target_size_one_million_columns():
    """north_star's target size -- NL >= 70 % of the HBM peak at NGPTOT >= 1 M columns *with the TL Taylor and AD symmetry tests passing
    on the GPU* -- is the one the bench times without looking at the results: 1 048 576 columns x NPROMA 128 is a 41 GB state, every
    sweep runs its 64-bit-offset variant (the arrays pass 4 GiB) and the block index passes 8191.  Here the results: NL bit-periodic
    over ALL columns and equal to the reference's SATUR + CLOUDSC2 on blocks 0, 1, 4096, 8191; TL and AD the same against CLOUDSC2TL /
    CLOUDSC2AD, the adjoint identity per column; and the two self-tests on a resident state of that size with the verdicts the
    reference prints for the full set of distinct blocks (tests/golden/drivers.json, NPROMA 128 x 3200 columns)."""
    import json

    import torch

    from tests.util import ROOT

    ngptot, nproma = 1048576, 128
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    assert ds.B_LOC.numel() * ds.B_LOC.element_size() > (1 << 32)  # past 32-bit byte offsets
    ds.nl(prm)
    torch.cuda.synchronize()
    outs = {"B_LOC_T": ds.B_LOC[:, 0], "B_LOC_Q": ds.B_LOC[:, 2], "B_LOC_QL": ds.B_LOC[:, 3], "B_LOC_QI": ds.B_LOC[:, 4], "PA": ds.PA,
            "PCOVPTOT": ds.PCOVPTOT, "PFPLSL": ds.PFPLSL, "PFPLSN": ds.PFPLSN, "PFHPSL": ds.PFHPSL, "PFHPSN": ds.PFHPSN}
    for n, t in outs.items():
        assert _periodic(t.permute(0, 2, 1).reshape(-1, t.shape[1])[:ngptot], ngptot), n
    assert not ds.B_LOC[:, 7].any()  # the driver's CLD(:,:,NCLV) = 0 plane
    chk = checker()
    set_lib_params(chk, prm)
    names = {"tent": "B_LOC_T", "tenq": "B_LOC_Q", "tenl": "B_LOC_QL", "teni": "B_LOC_QI", "clc": "PA", "covptot": "PCOVPTOT",
             "fplsl": "PFPLSL", "fplsn": "PFPLSN", "fhpsl": "PFHPSL", "fhpsn": "PFHPSN"}
    worst = 0.0
    for ibl in (0, 1, ds.nb // 2, ds.nb - 1):
        stb = c2.state_from_table(tab, nproma, nproma, col0=ibl * nproma)
        qs = chk.satur(np.ascontiguousarray(stb.PAP[0]), np.ascontiguousarray(stb.PT[0]), kfdia=nproma)
        ref = chk.cloudsc2(stb.ptsphy, refcall.block_inputs(stb, 0, qs), kfdia=nproma)
        for rn, dn in names.items():
            worst = max(worst, relerr(ref[rn], outs[dn][ibl].cpu().numpy()))
    print(f"1 048 576 columns, NL blocks 0, 1, {ds.nb // 2}, {ds.nb - 1} against SATUR + CLOUDSC2: worst {worst:.1e}")
    assert worst <= NL_TOL, worst
    del ds, outs
    torch.cuda.empty_cache()
    _full_size_tl_ad(ngptot, linearity=False)
    torch.cuda.empty_cache()
    # the two self-tests at that size, on a resident state
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "drivers.json")))["drivers"]
    rs = c2.ResidentState.from_table(tab, nproma, ngptot)
    zn, ok, itest, ms_t = rs.tl_taylor(c2.default_params(c2.ceta_from_table(tab), lregcl=False))
    ref = gold["tl_nproma128_ngptot3200"]
    assert np.allclose(zn[:6], ref["znormg"][:6], rtol=1e-6, atol=0), (zn, ref["znormg"])
    assert ok and ref["verdict"].startswith("TEST PASSED") and ref["verdict"].endswith(str(itest)), (ok, itest, ref["verdict"])
    zad, ok_ad, ms_a = rs.ad_symmetry(c2.default_params(c2.ceta_from_table(tab), lregcl=True))
    assert ok_ad and zad * np.finfo(np.float64).eps < 1e-12, zad
    print(f"1 048 576 columns resident: Taylor test passed (penalty {itest}, {ms_t:.1f} ms), adjoint test {zad:.1f} eps ({ms_a:.1f} ms)")


@pytest.mark.parametrize("nproma", [32, 64, 128, 256])
def test_full_size_nl_matches_the_reference_directly(nproma):
    """BASELINE configs[1] at its full size, compared DIRECTLY (every output, every column) with the checker's own driver:
    160 000 columns x 137 levels, state tiled on the device (cloudsc2_expand_launch) vs. the same state tiled on the host
    and run through CLOUDSC_DRIVER of the reference (cloudsc_driver_mod.F90:73-119, OpenMP over the blocks)."""
    import torch

    if not refcall.have_ref():
        pytest.skip("needs oracle/_ref (the reference driver) on the box")
    ngptot = 160000
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    want = c2.state_from_table(tab, nproma, ngptot)
    chk = refcall.RefLib()
    set_lib_params(chk, prm)
    chk.driver(0, 16, nproma, want.nlev, ngptot, want.ptsphy, want.driver_arrays())
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.nl(prm)
    torch.cuda.synchronize()
    got = ds.download(c2.state_from_table(tab, nproma, ngptot, poison_outputs=5.0))
    worst = {}
    for n, r in want.outputs().items():
        g = got.outputs()[n]
        assert np.all(np.isfinite(g)), n
        worst[n] = relerr(r, g)
        assert worst[n] <= NL_TOL, (n, worst[n])
    print("full-size NL, NPROMA", nproma, "worst relative difference per field:", {k: f"{v:.1e}" for k, v in worst.items()})


@pytest.mark.parametrize("flags", [dict(lregcl=True), dict()])
def test_tl_ad_16384_columns_match_checker_directly(flags):
    """BASELINE configs[3] size (NGPTOT = 16384): every TL output and every input adjoint of every column against the
    checker's CLOUDSC2TL / CLOUDSC2AD, block by block -- not through the adjoint identity."""
    nproma, ngptot = 128, 16384
    tab = c2.synthetic_table()
    r = _device_tl_ad(tab, nproma, ngptot, flags)
    worst_tl = max(relerr(r["tl_ref"][n], r["tl_dev"][n]) for n in r["tl_ref"])
    assert worst_tl <= TLAD_TOL, worst_tl
    assert_outputs_close(r["traj_ref"], r["traj_dev"], NL_TOL)
    worst_ad = 0.0
    for n in r["x_ref"]:
        ref_inc = r["x_ref"][n] - r["x0"][n] if n != "supsat" else r["x_ref"][n]
        got_inc = r["x_dev"][n] - r["x0"][n] if n != "supsat" else r["x_dev"][n]
        scale = max(np.abs(r["x_ref"][n]).max(), 1e-300)
        worst_ad = max(worst_ad, np.abs(got_inc - ref_inc).max() / scale)
    assert worst_ad <= TLAD_TOL, worst_ad
    print(f"16384 columns {flags}: worst TL {worst_tl:.1e}, worst AD {worst_ad:.1e}")


def test_taylor_test_in_fast_math():
    """The Taylor test evaluated with the FAST-math kernels (the ones bench.py times), requested per call through
    cloudsc2_params.math_mode = 1.  The verdict's V-shape rule (cloudsc_driver_tl_mod.F90:276-309) is decided by the
    round-off of the finite differences, which in fast math is as small as the reference's but not monotone; what must hold
    for a correct TL is the convergence itself: the ratio reaches 1 to better than 1e-5 and does so by lambda = 1e-4 at
    the latest (ISTART <= 4, :279-283).  The ratios, ISTART and the penalty are printed (DESIGN.md 4 quotes them)."""
    tab = c2.synthetic_table()
    for nproma in (32, 1):
        prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
        prm.math_mode = 1
        zn, ok, itest, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, 100), "tl")
        prm.math_mode = 2
        znp, okp, itestp, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, 100), "tl")
        z = np.abs(1.0 - zn)
        istart = int(np.argmax(z < 0.5)) + 1 if np.any(z < 0.5) else 0
        print(f"NPROMA {nproma} fast   : ratios-1 {np.array2string(zn - 1.0, precision=2)} ISTART {istart} penalty {itest} passed {ok}")
        print(f"NPROMA {nproma} precise: ratios-1 {np.array2string(znp - 1.0, precision=2)} penalty {itestp} passed {okp}")
        assert np.all(np.isfinite(zn))
        assert 1 <= istart <= 4, (nproma, zn)
        assert z.min() < 1e-5, (nproma, zn)
        assert np.allclose(zn[:6], znp[:6], rtol=1e-6, atol=0), (zn, znp)  # the two arithmetics agree where the test is not noise
        assert okp


def test_math_mode_is_per_call():
    """cloudsc2_params.math_mode selects the arithmetic of ONE call; the process default is only read.  Fast and precise
    results differ in the last bits, and a precise call in between must not change what a default-mode call returns."""
    import torch

    tab = c2.random_table(137, 100, seed=21)
    ds = c2.DeviceState.from_table(tab, 64, 1000)

    def run(mode):
        prm = c2.default_params(c2.ceta_from_table(tab))
        prm.math_mode = mode
        ds.nl(prm)
        torch.cuda.synchronize()
        return ds.PFPLSN.clone(), ds.B_LOC.clone()

    default_is_precise = B.get_math_mode()
    f0 = run(1)
    p = run(2)
    f1 = run(1)
    d = run(0)
    assert B.get_math_mode() == default_is_precise          # nothing process-wide was changed by the calls
    assert torch.equal(f0[0], f1[0]) and torch.equal(f0[1], f1[1])
    assert not torch.equal(f0[0], p[0])                      # the two arithmetics differ in the last bits ...
    assert relerr(p[0].cpu().numpy(), f0[0].cpu().numpy()) < 1e-12  # ... and only there
    want = p if default_is_precise else f0
    assert torch.equal(d[0], want[0]) and torch.equal(d[1], want[1])
    prm = c2.default_params(c2.ceta_from_table(tab))
    prm.math_mode = 3
    with pytest.raises(c2.Cloudsc2Error):
        ds.nl(prm)


def test_offset_variants_give_the_same_bits(tmp_path):
    """The launchers pick 32-bit byte offsets (C2F_OFF32) whenever every buffer is below 4 GiB, i.e. in every other test
    of this file; CLOUDSC2_OFF32=0 forces the 64-bit variants that large states use.  Both must produce identical bits
    (the choice is read once per process, hence the two child processes)."""
    import os
    import subprocess
    import sys

    from tests.util import ROOT

    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import dwarf_p_cloudsc2_tl_ad_amd as c2\n"
        "tab = c2.random_table(137, 100, seed=4)\n"
        "prm = c2.default_params(c2.ceta_from_table(tab))\n"
        "ds = c2.DeviceState.from_table(tab, 64, 1000)\n"
        "ds.nl(prm); ds.satur(prm)\n"
        "dx = ds.increments(); dy = c2.FlatFields('out', ds.nb, ds.nlev, ds.nproma, ds.device)\n"
        "ds.tl(prm, dx, dy); torch.cuda.synchronize()\n"
        "out = {n: getattr(ds, n).cpu().numpy() for n in ('B_LOC', 'PA', 'PFPLSL', 'PFPLSN', 'PFHPSL', 'PFHPSN')}\n"
        "out.update({'tl_' + n: t.cpu().numpy() for n, t in dy.t.items()})\n"
        "xa = c2.FlatFields('in', ds.nb, ds.nlev, ds.nproma, ds.device); sc = ds.new_scratch()\n"
        "ds.ad(prm, xa, dy, sc); torch.cuda.synchronize()\n"
        "out.update({'ad_' + n: t.cpu().numpy() for n, t in xa.t.items()})\n"
        "np.savez(sys.argv[1], **out)\n" % ROOT)
    files = []
    for mode in ("0", "1"):
        f = str(tmp_path / f"off32_{mode}.npz")
        r = subprocess.run([sys.executable, "-c", code, f], env={**os.environ, "CLOUDSC2_OFF32": mode}, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        files.append(np.load(f))
    assert set(files[0].files) == set(files[1].files) and len(files[0].files) == 32
    for n in files[0].files:
        assert np.array_equal(files[0][n], files[1][n]), n
    assert np.any(files[0]["tl_tent"] != 0.0) and np.any(files[0]["ad_t"] != 0.0)


def test_results_do_not_depend_on_the_blocking():
    """A column's result must not depend on NPROMA (which block and which lane it lands in): NL, TL and AD outputs per
    global column are compared bit for bit across awkward block sizes, ragged tails included."""
    import torch

    tab = c2.random_table(137, 100, seed=13)
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    ngptot = 3001

    def run(nproma):
        ds = c2.DeviceState.from_table(tab, nproma, ngptot)
        ds.nl(prm)
        ds.satur(prm)
        dx = ds.increments(zero_supsat=True)
        dy = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
        ds.tl(prm, dx, dy)
        y = {n: t.clone() for n, t in dy.t.items()}
        xa = c2.FlatFields("in", ds.nb, ds.nlev, ds.nproma, ds.device)
        ds.ad(prm, xa, dy, ds.new_scratch())
        torch.cuda.synchronize()

        def cols(t):  # (NBLOCKS, NLEVx, NPROMA) -> (column, level)
            return t.permute(0, 2, 1).reshape(-1, t.shape[1])[:ngptot].cpu().numpy()

        out = {"nl_" + n: cols(getattr(ds, n)) for n in ("PA", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")}
        out.update({f"nl_bloc{p}": cols(ds.B_LOC[:, p]) for p in (0, 2, 3, 4)})
        out.update({"tl_" + n: cols(t) for n, t in y.items()})
        out.update({"ad_" + n: cols(t) for n, t in xa.t.items()})
        return out

    base = run(128)
    assert np.any(base["ad_t"] != 0.0) and np.any(base["tl_tent"] != 0.0)
    for nproma in (1, 7, 33, 63, 65, 96, 127, 129, 257, 1000, 4000):
        got = run(nproma)
        for n in base:
            assert np.array_equal(base[n], got[n]), (nproma, n)


def test_strided_and_flat_layouts_agree():
    """The kernel-level ABI accepts any block stride per layout group: AoSoA planes (driver layout) and flat arrays
    must give identical results."""
    import torch

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, 64, 200)
    ds = c2.DeviceState(st, "cuda:0")
    ds.nl(prm)
    flat = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.nl(prm, outputs=flat.block())
    torch.cuda.synchronize()
    assert torch.equal(flat.t["tent"], ds.B_LOC[:, 0])
    assert torch.equal(flat.t["teni"], ds.B_LOC[:, 4])
    assert torch.equal(flat.t["fplsn"], ds.PFPLSN)
    assert torch.equal(flat.t["clc"], ds.PA)


def test_invalid_arguments_fail_loudly():
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, 32, 64)
    bad = c2.default_params(c2.ceta_from_table(tab)[:100])
    with pytest.raises(c2.Cloudsc2Error) as e:
        c2.run_state(bad, st, "nl")
    assert e.value.code == B.CLOUDSC2_EINVAL


@pytest.mark.parametrize("mode", [1, 2])
def test_nl_without_lphylin(mode):
    """YREPHLI%LPHYLIN=.false. -- the FOEALFA / FOEEWM form of the NL sweep's saturation pressure (cloudsc2.F90:349,365-369), off in
    every reference main -- against the reference Fortran with the same switch; SATUR stays in its LDPHYLIN form as in the driver
    (cloudsc_driver_mod.F90:91)."""
    tab = c2.random_table(137, 100, seed=17)
    prm = c2.default_params(c2.ceta_from_table(tab))
    prm.lphylin = 0
    prm.math_mode = mode
    st = c2.state_from_table(tab, 64, 300)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    assert_outputs_close(want, got, NL_TOL if mode == 2 else 10 * NL_TOL)
    prm.lphylin = 1
    lin = st.copy()
    c2.run_state(prm, lin, "nl")
    assert not np.array_equal(lin.B_LOC, got.B_LOC)  # the switch is read


def _run_fortran(exe, *args, cwd=None, env=None):
    import os
    import subprocess

    from tests.util import ROOT

    path = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran", "build", exe)
    if not os.path.exists(path):
        pytest.fail(f"{path} missing: run __graft_entry__.build()")
    full = None if env is None else {k: v for k, v in {**os.environ, **env}.items() if v != ""}  # (a variable set to "" is removed)
    r = subprocess.run([path, *map(str, args)], capture_output=True, text=True, timeout=300, cwd=cwd, env=full)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout, r.stderr


def test_fortran_drivers_through_iso_c_binding():
    """The Fortran drivers with the reference's signatures (fortran/cloudsc_driver*_mod.F90) call the same C ABI:
    dwarf-cloudsc2-nl must reproduce the Python driver's outputs, -tl and -ad must print the reference's verdicts
    for the README invocations (README.md:47-62: `nl 4 160000 32`, `tl 1 100 1`, `ad 1 100 100`)."""
    import re

    out, err = _run_fortran("dwarf-cloudsc2-nl", 4, 16000, 32)
    assert "NGPBLKS=500" in err
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    st = c2.state_from_table(tab, 32, 16000)
    c2.run_state(prm, st, "nl")
    want = {"PFPLSN": st.PFPLSN, "PFHPSN": st.PFHPSN, "PFPLSL": st.PFPLSL, "PA": st.PA, "TENDENCY_T": st.B_LOC[:, 0],
            "TENDENCY_Q": st.B_LOC[:, 2], "TENDENCY_L": st.B_LOC[:, 3], "TENDENCY_I": st.B_LOC[:, 4]}
    for name, a in want.items():
        m = re.search(rf"^\s*{name}\s+(\S+)\s+(\S+)\s+(\S+)\s*$", out, flags=re.M)
        assert m, (name, out)
        mn, mx, s = map(float, m.groups())
        assert abs(mn - a.min()) <= 1e-12 * max(1e-300, np.abs(a).max())
        assert abs(mx - a.max()) <= 1e-12 * max(1e-300, np.abs(a).max())
        assert abs(s - np.abs(a).sum()) <= 1e-9 * np.abs(a).sum()  # Fortran SUM is sequential over 2e6 terms

    # the default main runs the sweep on a resident state first, then the reference flow, and prints the two rates side by side
    assert err.count("NGPBLKS=500") == 2 and "(state resident on the GPU)" in err, err
    m1 = re.search(r"state resident on the GPU \(cloudsc2_state_\*, CLOUDSC2_RESIDENT=1\):\s+([0-9.]+) ms =\s+([0-9.E+]+) columns/s per sweep", err)
    m2 = re.search(r"CLOUDSC_DRIVER on host arrays \(the reference flow, PCIe-bound\):\s+([0-9.]+) ms =\s+([0-9.E+]+) columns/s, first call", err)
    assert m1 and m2, err
    assert float(m1.group(2)) > 5.0 * float(m2.group(2)), (m1.group(0), m2.group(0))  # (16 000 columns: the link costs far more than the sweep)
    out0, err0 = _run_fortran("dwarf-cloudsc2-nl", 4, 16000, 32, env={"CLOUDSC2_RESIDENT": "0"})  # the reference-identical flow alone
    assert err0.count("NGPBLKS=500") == 1 and "resident" not in err0, err0
    summary = lambda o: [ln for ln in o.splitlines() if re.match(r"^\s*(P[A-Z]+|TENDENCY_[A-Z])\s", ln)]  # noqa: E731
    assert summary(out0) == summary(out) and len(summary(out)) >= 8, (out0, out)

    # the self-tests: resident by default, the host-array drivers (CLOUDSC_DRIVER_TL / _AD through ISO_C_BINDING) with CLOUDSC2_RESIDENT=0
    for env in (None, {"CLOUDSC2_RESIDENT": "0"}):
        out, err = _run_fortran("dwarf-cloudsc2-tl", 1, 100, 1, env=env)
        assert "TEST PASSED, penalty" in out, out
        assert ("(state resident on the GPU)" in err) == (env is None), err
        out, err = _run_fortran("dwarf-cloudsc2-ad", 1, 100, 100, env=env)
        assert "TEST OK" in out, out
        assert ("(state resident on the GPU)" in err) == (env is None), err


@pytest.mark.parametrize("nproma, ngptot, nproma_stat, mode, levapls2", [(128, 1000, 128, 2, False), (32, 333, 32, 1, False),
                                                                          (100, 250, 100, 2, True), (128, 777, 16, 1, False),
                                                                          (1, 61, 1, 2, False), (128, 3000, 1300, 1, False),
                                                                          (2048, 2500, 2048, 1, False)])
def test_lambda_sweep_equals_ten_perturbed_runs(nproma, ngptot, nproma_stat, mode, levapls2):
    """cloudsc2_taylor_sweep_launch (the ten lambdas on the lanes of a wave, nothing stored) against what it replaces: ten
    perturbed NL launches that store their outputs, each followed by cloudsc2_taylor_sums_launch
    (cloudsc_driver_tl_mod.F90:197-244).  Both evaluate the same level function on the same perturbed inputs, so a term
    F - F5 differs by rounding only; a block's sum is compared against the sum of the terms' magnitudes.  The denominators
    (sums of the TL outputs) differ by summation order only."""
    import torch

    from dwarf_p_cloudsc2_tl_ad_amd.state import PLANE_Q, PLANE_QI, PLANE_QL, PLANE_T

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False, levapls2=levapls2)
    prm.math_mode = mode
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.satur(prm)
    ds.nl(prm, fused_satur=False)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    # a second set of output arrays for the perturbed runs, laid out like the state's own (what the NL launch requires)
    from dwarf_p_cloudsc2_tl_ad_amd.driver import _fld

    pc = {n: torch.zeros_like(getattr(ds, n)) for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")}
    S, H = ds.nproma * ds.nlev, ds.nproma * (ds.nlev + 1)
    po = B.Outputs()
    po.tent, po.tenq = _fld(pc["B_LOC"], PLANE_T * S, 8 * S), _fld(pc["B_LOC"], PLANE_Q * S, 8 * S)
    po.tenl, po.teni = _fld(pc["B_LOC"], PLANE_QL * S, 8 * S), _fld(pc["B_LOC"], PLANE_QI * S, 8 * S)
    po.clc, po.covptot = _fld(pc["PA"], 0, S), _fld(pc["PCOVPTOT"], 0, S)
    po.fplsl, po.fplsn = _fld(pc["PFPLSL"], 0, H), _fld(pc["PFPLSN"], 0, H)
    po.fhpsl, po.fhpsn = _fld(pc["PFHPSL"], 0, H), _fld(pc["PFHPSN"], 0, H)
    ds.increments(into=inc)
    ds.tl(prm, inc, dout, store_traj=False)
    sweep = ds.taylor_sweep(prm, dout, nproma_stat=nproma_stat)
    torch.cuda.synchronize()
    assert torch.isfinite(sweep).all()
    # the ten compared fields in the order of the ERROR_NORM calls (cloudsc_driver_tl_mod.F90:233-242), and their size
    fields = (("tent", ds.B_LOC[:, PLANE_T]), ("tenq", ds.B_LOC[:, PLANE_Q]), ("tenl", ds.B_LOC[:, PLANE_QL]), ("teni", ds.B_LOC[:, PLANE_QI]),
              ("clc", ds.PA), ("fplsl", ds.PFPLSL), ("fplsn", ds.PFPLSN), ("fhpsl", ds.PFHPSL), ("fhpsn", ds.PFHPSN), ("covptot", ds.PCOVPTOT))
    size = [float(t.abs().max()) for _, t in fields]
    names = [n for n, _ in fields]
    nbs = (ngptot + nproma_stat - 1) // nproma_stat
    for il in range(10):
        lam = 10.0 ** -(il + 1)
        ds.nl(prm, fused_satur=False, pert_lambda=lam, outputs=po)
        if nproma_stat == nproma:
            old = ds.taylor_sums(po, dout, lam)
            torch.cuda.synchronize()
            got, ref = sweep[il].cpu().numpy(), old.cpu().numpy()
            den_tol = 1e-11 * np.abs(ref[:, :, 1]) + 1e-12 * np.abs(ref[:, :, 1]).max(axis=0, keepdims=True) + 1e-300
            assert np.all(np.abs(got[:, :, 1] - ref[:, :, 1]) <= den_tol), (il, np.abs(got[:, :, 1] - ref[:, :, 1]).max())
            # numerators: per field, against the magnitude of what is summed (the base field's own size x eps x terms)
            for f, nm in enumerate(names):
                scale = size[f] * ds.nlev * nproma
                assert np.all(np.abs(got[:, f, 0] - ref[:, f, 0]) <= 64 * np.finfo(np.float64).eps * scale + 1e-300), (il, nm)
        else:
            torch.cuda.synchronize()
            # another block of the statistic than the arrays' blocking: the block sums must add up to the same totals
            old = ds.taylor_sums(po, dout, lam)
            torch.cuda.synchronize()
            tot_new, tot_old = sweep[il].sum(dim=0).cpu().numpy(), old.sum(dim=0).cpu().numpy()
            assert sweep[il].shape[0] == nbs
            assert np.allclose(tot_new[:, 1], tot_old[:, 1], rtol=1e-10, atol=1e-12 * np.abs(tot_old[:, 1]).max() + 1e-300)
            for f, nm in enumerate(names):
                scale = size[f] * ds.nlev * ngptot
                assert abs(tot_new[f, 0] - tot_old[f, 0]) <= 64 * np.finfo(np.float64).eps * scale + 1e-300, (il, nm)


@pytest.mark.parametrize("nproma, ngptot, mode, levapls2, sup", [(128, 1000, 1, False, 0.01), (32, 333, 2, False, 0.0), (100, 250, 1, True, 0.01)])
def test_tl_with_increments_formed_in_the_sweep(nproma, ngptot, mode, levapls2, sup):
    """cloudsc2_tl_launch_self (dx = 0.01*x formed inside the TL sweep, what both test drivers use) against cloudsc2_tl_launch fed
    with the same increments from memory (cloudsc_driver_tl_mod.F90:156-171; cloudsc_driver_ad_mod.F90:124-139 with ZSUPSAT = 0).
    The same operations on the same values; the compiler may contract 0.01*x into a following add, hence a few ulp."""
    import torch

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False, levapls2=levapls2)
    prm.math_mode = mode
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.PSUPSAT.uniform_(0.0, 1e-4)  # (zero in the synthetic table: make the PSUPSAT increment matter)
    ds.satur(prm)
    inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    _, dself = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    ds.increments(zero_supsat=(sup == 0.0), into=inc)
    ds.tl(prm, inc, dout)
    traj = {n: getattr(ds, n).clone() for n in ("B_LOC", "PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN")}
    ds.tl(prm, None, dself, supsat_increment=sup)
    torch.cuda.synchronize()
    for n, t in traj.items():
        assert torch.equal(getattr(ds, n), t), n  # the trajectory outputs do not depend on where the increments come from
    for n in B.OUT_NAMES:
        a, b = dout.t[n], dself.t[n]
        assert torch.isfinite(b).all()
        scale = float(a.abs().max())
        floor = 1e-15 if n in ("clc", "covptot") else 1e-300  # (cover perturbations that are cancellation noise of 1e-18 themselves)
        assert float((a - b).abs().max()) <= 1e-13 * scale + floor, (n, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("nproma, ngptot, mode", [(128, 1000, 1), (32, 333, 2), (100, 250, 1)])
def test_adjoint_norms_formed_in_the_sweeps(nproma, ngptot, mode):
    """The adjoint test's norms as its driver now forms them -- <y,y> in the TL sweep (cloudsc2_tl_launch_self's yy), <x0,x_adj> and
    norm3 in the reverse sweep (cloudsc2_ad_launch_reverse_norms) -- against cloudsc2_adjoint_norms_launch over the stored arrays
    (cloudsc_driver_ad_mod.F90:184-195,240-264)."""
    import torch

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    prm.math_mode = mode
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.satur(prm)
    x, y = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, ds.device)
    ncp = ds.nb * ds.nproma
    f64 = dict(dtype=torch.float64, device=ds.device)
    na, nb_ = torch.zeros((3, ncp), **f64), torch.zeros((3, ncp), **f64)
    ga, gb = torch.zeros(1, **f64), torch.zeros(1, **f64)
    act = torch.arange(ncp, device=ds.device) < ngptot

    ds.tl(prm, None, y, supsat_increment=0.0, yy=na)
    ds.adjoint_norms(y, None, nb_, gb)
    ds.ad_reverse_norms(prm, x, y, na, ga)
    torch.cuda.synchronize()
    xa = {n: t.clone() for n, t in x.t.items()}
    assert all(float(t.abs().max()) == 0.0 for t in y.t.values())  # the output adjoints are consumed

    ds.tl(prm, None, y, supsat_increment=0.0)
    ds.ad(prm, x, y, None, assign=True, sweep="reverse")
    ds.adjoint_norms(None, x, nb_, gb)
    torch.cuda.synchronize()
    for n, t in x.t.items():
        scale = float(t.abs().max())
        assert float((t - xa[n]).abs().max()) <= 1e-13 * scale, n
    a, b = na[:, act].cpu().numpy(), nb_[:, act].cpu().numpy()
    assert np.allclose(a[0], b[0], rtol=1e-13, atol=0)
    assert np.allclose(a[1], b[1], rtol=1e-12, atol=0), np.abs(a[1] / b[1] - 1).max()
    eps = np.finfo(np.float64).eps
    assert np.array_equal(a[2], np.abs(a[0] - a[1]) / eps / a[1])
    assert float(ga) == np.abs(a[2]).max() and float(gb) == np.abs(b[2]).max()
    assert float(ga) < 1e4 and float(gb) < 1e4


@pytest.mark.parametrize("flags, full", [(dict(lregcl=True, levapls2=True), "0"), (dict(lregcl=True), "1"), (dict(lregcl=True), "0")])
def test_adjoint_test_driver_in_both_sequences(flags, full, monkeypatch):
    """cloudsc2_ad_symmetry_run's two sequences: the fused one (TL forming <y,y>, reverse sweep forming the other norms) and the plain
    one -- TL, norm-1 kernel, both sweeps of CLOUDSC2AD, norm-2/3 kernel -- which the evaporation branch requires (its cover checkpoints
    come from the forward sweep) and CLOUDSC2_AD_SYMMETRY_FULL=1 selects for measurements.  The identity must hold in all of them."""
    monkeypatch.setenv("CLOUDSC2_AD_SYMMETRY_FULL", full)
    tab = c2.random_table(137, 100, seed=11) if flags.get("levapls2") else c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, 64, 300)
    zad, ok, _ = c2.run_state(prm, st, "ad")
    assert np.isfinite(zad) and ok, (flags, full, zad)
    assert zad * np.finfo(np.float64).eps < 1e-12, zad


@pytest.mark.parametrize("nproma, ngptot", [(16, 80), (50, 130), (128, 300)])
def test_test_drivers_against_the_reference_drivers_run_here(nproma, ngptot):
    """The reference's own CLOUDSC_DRIVER_TL and CLOUDSC_DRIVER_AD (oracle/_ref, compiled from the reference's sources) run on this
    box on the same state, at blockings and sizes other than the golden file's: the ten Taylor ratios agree where they are not
    finite-difference noise (lambda >= 1e-6) and the verdict with its penalty is the same; the adjoint test passes in both with a
    maximum error of a few epsilon."""
    import re
    import sys

    if not refcall.have_ref():
        pytest.skip("needs oracle/_ref (the reference drivers) on the box")

    def capture_stdout(fn):
        sys.stdout.flush()
        return refcall.RefLib._capture(1, fn)
    tab = c2.synthetic_table()
    ref = refcall.RefLib()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    set_lib_params(ref, prm)
    st = c2.state_from_table(tab, nproma, ngptot)
    txt = capture_stdout(lambda: ref.driver(1, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))
    want = np.array([float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", txt, flags=re.M)][:10])
    verdict = re.search(r"TEST (PASSED|FAILLED).*", txt).group(0).strip()
    zn, ok, itest, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, ngptot), "tl")
    assert want.shape == (10,), txt
    assert np.allclose(zn[:6], want[:6], rtol=1e-6, atol=0), (zn, want)
    assert ok == verdict.startswith("TEST PASSED") and verdict.endswith(str(itest)), (verdict, ok, itest)

    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    set_lib_params(ref, prm)
    st = c2.state_from_table(tab, nproma, ngptot)
    txt = capture_stdout(lambda: ref.driver(2, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))
    err = float(re.search(r"maximum error is\s+([0-9.Ee+-]+)", txt).group(1))
    zad, ok, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, ngptot), "ad")
    assert "TEST OK" in txt and ok
    assert err < 100 and zad < 100, (err, zad)  # both a few epsilon (the reference's threshold is 1e4)


def test_test_drivers_with_the_evaporation_branch_against_the_reference_drivers():
    """LEVAPLS2 = .TRUE. (off in every shipped main) through the reference's own CLOUDSC_DRIVER_TL and CLOUDSC_DRIVER_AD run here
    (oracle/_ref) and through ours: the Taylor ratios agree where they are not finite-difference noise -- the tangent block of the
    branch is evaluated in the log-derivative form since round 5 (level_tl: 0.5777 ZBETA5 dB / B, no pow) against the reference's
    cloudsc2tl.F90:871-882 --, the ratios converge to 1, and the adjoint test passes in both with a maximum error of a few epsilon."""
    import re
    import sys

    if not refcall.have_ref():
        pytest.skip("needs oracle/_ref (the reference drivers) on the box")
    tab = c2.random_table(137, 100, seed=5)   # (non-zero PSUPSAT and cloud tendencies; the branch is exercised in most columns)
    ref = refcall.RefLib()
    nproma, ngptot = 50, 130
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=False, levapls2=True)
    set_lib_params(ref, prm)
    st = c2.state_from_table(tab, nproma, ngptot)
    sys.stdout.flush()
    txt = refcall.RefLib._capture(1, lambda: ref.driver(1, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))
    want = np.array([float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", txt, flags=re.M)][:10])
    zn, ok, itest, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, ngptot), "tl")
    print("Taylor ratios with LEVAPLS2: reference", want, "| GPU", zn, ok, itest)
    assert want.shape == (10,), txt
    verdict = re.search(r"TEST (PASSED|FAILLED).*", txt).group(0).strip()
    # with the branch on the increments of lambda >= 1e-4 cross its clip / reset discontinuities (ratios 49 ... 10 907 in the reference
    # itself: `TEST FAILLED, err 13`); from lambda = 1e-5 on the ratio is 1 + 1.6e-6, 1 + 1.6e-7, ...: the tangent IS the derivative
    assert verdict.startswith("TEST FAILLED") and verdict.split()[-1] == "13" and not ok and itest == 13, (verdict, ok, itest)
    assert np.allclose(zn[:4], want[:4], rtol=1e-5, atol=0), (zn, want)  # (measured: equal to 1e-6, discontinuities and all)
    assert np.allclose(zn[4:6], want[4:6], rtol=1e-6, atol=0), (zn, want)
    assert abs(zn[5] - 1.0) < abs(zn[4] - 1.0) < 1e-5
    got = c2.state_from_table(tab, nproma, ngptot)
    c2.run_state(prm, got, "nl")
    assert np.any(got.PCOVPTOT != 0.0)  # the branch really ran

    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True, levapls2=True)
    set_lib_params(ref, prm)
    st = c2.state_from_table(tab, nproma, ngptot)
    txt = refcall.RefLib._capture(1, lambda: ref.driver(2, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))
    err = float(re.search(r"maximum error is\s+([0-9.Ee+-]+)", txt).group(1))
    zad, ok_ad, _ = c2.run_state(prm, c2.state_from_table(tab, nproma, ngptot), "ad")
    print("adjoint test with LEVAPLS2: reference", err, "eps | GPU", zad, "eps")
    assert "TEST OK" in txt and ok_ad
    assert err < 200 and zad < 200, (err, zad)  # both a few epsilon (the reference's threshold is 1e4)


def test_pacing_of_partial_rounds_changes_no_bits():
    """TL / AD launches of a few partial rounds of workgroups are paced (cloudsc2_column.hpp: struct Pace -- workgroups whose slot has
    one workgroup less to run nap at every level; 140 000 columns = 1094 workgroups on 512 slots: 2 rounds + 70): a matter of WHEN
    waves run, never of what they compute -- like the nap of the lighter SIMDs in a one-round NL launch.  Fresh processes with both on
    (default) and off give identical bits for every NL output, every TL output and every input adjoint; the launcher reports the
    pacing for this size and not for 16 384 columns."""
    import subprocess
    import sys

    from tests.util import ROOT

    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "import torch, dwarf_p_cloudsc2_tl_ad_amd as c2\n"
        "tab = c2.synthetic_table(); prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True)\n"
        "for n in (16384, 140000):\n"
        "    ds = c2.DeviceState.from_table(tab, 128, n); h = hashlib.sha256()\n"
        "    ds.nl(prm); torch.cuda.synchronize()\n"   # (one round of waves at both sizes: the lighter SIMDs of the fullest CUs nap)
        "    for t in (ds.B_LOC, ds.PA, ds.PCOVPTOT, ds.PFPLSL, ds.PFPLSN, ds.PFHPSL, ds.PFHPSN): h.update(t.cpu().numpy().tobytes())\n"
        "    ds.satur(prm)\n"
        "    dx = ds.increments(zero_supsat=True); dy = c2.FlatFields('out', ds.nb, ds.nlev, ds.nproma, ds.device)\n"
        "    ds.tl(prm, dx, dy); torch.cuda.synchronize()\n"
        "    for k in sorted(dy.t): h.update(dy.t[k].cpu().numpy().tobytes())\n"
        "    xa = c2.FlatFields('in', ds.nb, ds.nlev, ds.nproma, ds.device)\n"
        "    ds.ad(prm, xa, dy, ds.new_scratch()); torch.cuda.synchronize()\n"
        "    for k in sorted(xa.t): h.update(xa.t[k].cpu().numpy().tobytes())\n"
        "    print('HASH', n, h.hexdigest())\n" % ROOT)

    def run(env):
        e = {k: v for k, v in os.environ.items() if not k.startswith(("CLOUDSC2_PACE", "CLOUDSC2_NL_LIGHT"))}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env={**e, **env})
        assert r.returncode == 0, r.stderr[-3000:]
        return [ln for ln in r.stdout.splitlines() if ln.startswith("HASH")], r.stderr

    on, err_on = run({"CLOUDSC2_PACE_VERBOSE": "1"})
    off, err_off = run({"CLOUDSC2_PACE": "0", "CLOUDSC2_NL_LIGHT": "0", "CLOUDSC2_PACE_VERBOSE": "1"})
    assert len(on) == 2 and on == off, (on, off)
    assert "launch of 1094 workgroups on 512 slots paced: 2 whole rounds + 70 workgroups" in err_on, err_on[-1500:]
    assert "launch of 128 workgroups" not in err_on and "paced" not in err_off


def test_the_dispatch_rule_holds_on_this_device():
    """The nap of the lighter SIMDs in one-round NL launches rests on a rule about where the dispatcher puts the waves of a launch
    (cloudsc2_simd_population).  The library checks it on the device before using it (a 40 us probe launch whose waves record their
    HW_ID; one miss and the nap stays off); the same probe through the ABI: every wave sits on a SIMD with the predicted number of waves."""
    import ctypes as C

    import torch

    torch.cuda.synchronize()  # (an idle device: other work would be placed between the probe's waves)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    wrongs = []
    for _ in range(3):
        checked, wrong = C.c_longlong(), C.c_longlong()
        B.check(B.lib.cloudsc2_dispatch_probe(C.byref(checked), C.byref(wrong)))
        assert checked.value == 2 * (5 * cus - cus // 8 - 2)
        wrongs.append(wrong.value)
    print("dispatch probe: waves off the rule in three probes:", wrongs)
    # measured 0 / 0 / 0 on every box; one disturbed probe (another process touching the GPU) is tolerated here -- the library itself
    # then simply leaves the nap off -- a rule that is wrong shows in all three
    assert min(wrongs) == 0 and sorted(wrongs)[1] == 0, wrongs


def test_the_pace_rule_holds_on_this_device():
    """TL / AD pacing rests on a rule about the ORDER in which the dispatcher hands workgroups to freed slots (Pace::begin decides
    from blockIdx mod slots alone which workgroups sit on the launch's critical path).  The library checks it on the device at a
    synchronous moment before it ever paces a launch (cloudsc2_device_prepare: a 130 us probe of 2.44 rounds of workgroups that only
    stay as long as their class would, and record where and when they ran; one miss and pacing stays off); here the same probe
    through the ABI, for the occupancy the fp64 TL / AD kernels really have, and the cached verdicts the launchers read."""
    import ctypes as C

    import torch

    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    per_cu = C.c_int()
    occ = {}
    for kernel, flags, name in ((1, 1 | 8 | 32, "tl"), (2, 1 | 32, "ad"), (3, 1 | 32, "ad_reverse")):  # C2F_QSAT [| C2F_TRAJ] | C2F_OFF32
        B.check(B.lib.cloudsc2_kernel_occupancy(kernel, flags, C.byref(per_cu)))
        occ[name] = per_cu.value
    print("workgroups per CU:", occ)
    if not B.SINGLE:
        assert set(occ.values()) == {2}, occ  # one wave per SIMD: 311 / 308 registers
    for pc in sorted(set(occ.values())):
        wrongs = []
        for _ in range(3):
            checked, wrong = C.c_longlong(), C.c_longlong()
            B.check(B.lib.cloudsc2_pace_probe(pc, C.byref(checked), C.byref(wrong)))
            slots = cus * pc
            assert checked.value == 2 * slots + slots * 113 // 256
            wrongs.append(wrong.value)
        print(f"pace probe, {pc} workgroup(s) per CU: workgroups off the rule in three probes:", wrongs)
        assert min(wrongs) == 0 and sorted(wrongs)[1] == 0, wrongs  # (one disturbed probe tolerated, as for the dispatch rule)
    # the synchronous moment itself, and what the launchers will read afterwards
    B.check(B.lib.cloudsc2_device_prepare())
    nap, pace = C.c_int(-2), C.c_int(-2)
    B.check(B.lib.cloudsc2_device_rules(occ["tl"], C.byref(nap), C.byref(pace)))
    print("cached verdicts: NL nap", nap.value, "TL/AD pacing", pace.value)
    assert nap.value in (0, 1) and pace.value in (0, 1)  # probed (an undisturbed box gives 1 / 1)
    B.check(B.lib.cloudsc2_device_rules(7, C.byref(nap), C.byref(pace)))
    assert pace.value == -1  # no kernel of the build has that occupancy: never probed, never paced
    assert B.lib.cloudsc2_kernel_occupancy(9, 0, C.byref(per_cu)) != 0


def test_launches_under_stream_capture_are_plain_kernel_nodes():
    """No launcher allocates, copies or synchronises on behalf of the launch heuristics (round 4's NL launcher ran its dispatch probe
    from the first one-round launch): an NL and a TL launch recorded into a graph while the stream is capturing -- the first launches
    of a FRESH process whose device was prepared by the state's allocation -- replay to the bits of the eager launches
    (tools/capture_probe.py), and the captured TL launch was paced from the cached verdict."""
    import subprocess
    import sys

    from tests.util import ROOT

    e = {k: v for k, v in os.environ.items() if not k.startswith(("CLOUDSC2_PACE", "CLOUDSC2_NL_LIGHT"))}
    for mode in ("cold",):  # (the captured launches are the first of their process -- the level table included; `warm` stays in the tool)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "capture_probe.py"), mode], capture_output=True, text=True, timeout=600,
                           env={**e, "CLOUDSC2_PACE_VERBOSE": "1"})
        assert r.returncode == 0 and f"CAPTURE OK {mode}" in r.stdout, (mode, r.stdout[-500:], r.stderr[-3000:])
        # the device was prepared by the allocation (both probes ran there, before any launch)
        assert "dispatch probe on device" in r.stderr and "pace probe on device" in r.stderr, r.stderr[-3000:]
        if "TL / AD pacing is on" in r.stderr:
            assert r.stderr.index("pace probe on device") < r.stderr.index("launch of 1094 workgroups on 512 slots paced"), r.stderr[-3000:]


def test_launchers_called_from_four_host_threads_at_once():
    """The reference's kernels are called concurrently from NUMOMP threads on disjoint blocks (cloudsc_driver_mod.F90:73-119); the
    counterpart here: four host threads, each with its own state and its own stream, launching NL, SATUR, TL and AD at once in a FRESH
    process -- so that the first-use paths race too: the level tables of two vertical grids, the occupancy caches, the pacing
    decisions, the per-call arithmetic mode (two threads fast, two precise).  Every thread's results equal, bit for bit, what the same
    calls give one after the other; and the host-array driver (one workspace, serialised by the library) survives two threads."""
    import subprocess
    import sys

    from tests.util import ROOT

    code = r'''
import sys, threading; sys.path.insert(0, %r)
import numpy as np, torch, dwarf_p_cloudsc2_tl_ad_amd as c2
specs = [(137, 128, 20000, 1), (91, 64, 9000, 2), (137, 32, 5000, 2), (91, 100, 12345, 1)]   # (nlev, nproma, ngptot, math mode)
def make(nlev, nproma, ngptot, mode):
    tab = c2.random_table(nlev, 60, seed=nlev + nproma)
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=True); prm.math_mode = mode
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    dx = c2.FlatFields("in", ds.nb, ds.nlev, ds.nproma, ds.device); dy = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, ds.device)
    xa = c2.FlatFields("in", ds.nb, ds.nlev, ds.nproma, ds.device)
    return prm, ds, dx, dy, xa
def work(item, stream, reps):
    prm, ds, dx, dy, xa = item
    for _ in range(reps):
        ds.nl(prm, stream); ds.satur(prm, stream); ds.increments(zero_supsat=True, into=dx)
        ds.tl(prm, dx, dy, stream); xa.zero_(); ds.ad(prm, xa, dy, ds.new_scratch(), stream)
def digest(item):
    prm, ds, dx, dy, xa = item
    torch.cuda.synchronize()
    return [t.clone() for t in (ds.B_LOC, ds.PA, ds.PCOVPTOT, ds.PFPLSL, ds.PFPLSN, ds.PFHPSL, ds.PFHPSN)] + [xa.t[k].clone() for k in sorted(xa.t)]
items = [make(*s) for s in specs]          # (allocation = the synchronous moment: the device is prepared; nothing has been launched yet)
streams = [torch.cuda.Stream() for _ in specs]
errs = []
def guarded(i):
    try:
        with torch.cuda.stream(streams[i]):
            work(items[i], streams[i], 5)
    except Exception as e:
        errs.append(repr(e))
th = [threading.Thread(target=guarded, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
assert not errs, errs
par = [digest(it) for it in items]
for i, it in enumerate(items):             # the same calls one after the other, on the default stream
    work(it, None, 1)
seq = [digest(it) for it in items]
for a, b in zip(par, seq):
    assert all(torch.equal(x, y) for x, y in zip(a, b))
# the host-array driver from two threads: one workspace, serialised inside the library
tab = c2.synthetic_table(); prm = c2.default_params(c2.ceta_from_table(tab))
sts = [c2.state_from_table(tab, 64, 3000), c2.state_from_table(tab, 128, 5000)]
th = [threading.Thread(target=lambda s=s: c2.run_state(prm, s, "nl")) for s in sts]
[t.start() for t in th]; [t.join() for t in th]
for s in sts:
    r = c2.state_from_table(tab, s.nproma, s.ngptot); c2.run_state(prm, r, "nl")
    assert all(np.array_equal(a, b) for a, b in zip(s.outputs().values(), r.outputs().values()))
print("THREADS OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "THREADS OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
