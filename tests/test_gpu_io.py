"""Device-side data-format kernels (include/cloudsc2_hip.h: cloudsc2_expand_launch, cloudsc2_validate_launch) against
host restatements of EXPAND_R2/R3 (expand_mod.F90:270-335) and VALIDATE_R2/R3 (validate_mod.F90:165-261), and the
file -> device state -> NL -> validation-report chain of the reference's main (dwarf_cloudsc.F90:79-124)."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from tests.util import ROOT, B, c2, refcall
from dwarf_p_cloudsc2_tl_ad_amd import fileio

pytestmark = pytest.mark.gpu


def host_validate(field, ref_tab, nproma, ngptot, start=0, period=None):
    """VALIDATE_R2 / R3 in numpy: field (NBLOCKS, [NDIM,] NLEVx, NPROMA), ref_tab ([NDIM,] NLEVx, KLON)."""
    klon = ref_tab.shape[-1]
    period = klon if period is None else period
    cols = (start + np.arange(ngptot) % period) % klon
    nlevx = field.shape[-2]
    if field.ndim == 3:
        f = field.transpose(0, 2, 1).reshape(-1, nlevx)[:ngptot]           # (col, lev)
        r = ref_tab.T[cols]
    else:
        ndim = field.shape[1]
        f = field.transpose(0, 3, 1, 2).reshape(-1, ndim, nlevx)[:ngptot]  # (col, dim, lev)
        r = ref_tab.transpose(2, 0, 1)[cols]
    d = np.abs(f - r)
    return np.array([field.min(), field.max(), d.max(), d.sum(), np.abs(r).sum()])


@pytest.mark.parametrize("nproma,ngptot,start,period", [(32, 250, 0, None), (100, 100, 0, None), (128, 1000, 37, None),
                                                        (64, 70, 25, 25), (1, 7, 0, 7)])
def test_device_expand_equals_host_tiling(nproma, ngptot, start, period):
    import torch

    tab = c2.random_table(137, 100, seed=11)
    host = c2.state_from_table(tab, nproma, ngptot, col0=start) if period is None else None
    ds = c2.DeviceState.from_table(tab, nproma, ngptot, start=start, period=period)
    torch.cuda.synchronize()
    if host is not None:
        for n in ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT", "B_CML", "PCLV"):
            assert np.array_equal(getattr(ds, n).cpu().numpy(), getattr(host, n)), n
        for n in ("PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "B_LOC"):
            assert not getattr(ds, n).any(), n
    else:
        # a rank whose table slice is shorter than KLON (GET_OFFSETS: size = min(nlon, ngptot), start = rank offset)
        cols = ds.PT.cpu().numpy().transpose(0, 2, 1).reshape(-1, 137)
        idx = (start + np.arange(ngptot) % period) % 100  # GET_OFFSETS' slice START..END tiled with period SIZE
        assert np.array_equal(cols[:ngptot], tab["PT"].T[idx]) and not cols[ngptot:].any()


def _fortran_real(tok: str) -> float:
    """E20.13 output back to a number ('0.1000000000000-299' has no E)."""
    import re

    return float(re.sub(r"(?<=\d)([+-]\d{3})$", r"E\1", tok))


@pytest.mark.skipif(not refcall.have_ref(), reason="oracle/_ref/libcloudsc2_ref.so did not travel")
@pytest.mark.parametrize("nproma,ngptot,split", [(32, 250, None), (128, 1000, None), (64, 70, None), (16, 25, (100, 2, 4)), (2, 13, (50, 1, 4))])
def test_device_tiler_and_validator_against_the_reference_modules(nproma, ngptot, split):
    """The device tiler = EXPAND_R2 / EXPAND_R3 of the reference's expand_mod (bit for bit, also for a rank of a multi-rank run:
    GET_OFFSETS + LOAD_AND_EXPAND), and the device validator's report = VALIDATE_R2 / R3 + ERROR_PRINT of validate_mod run on
    the host arrays: minimum, maximum and largest error equal in every printed digit, the two sums (whose order of addition
    differs) to 1e-12.  The modules are compiled unmodified into oracle/_ref (oracle/Makefile)."""
    import torch

    reflib = refcall.RefLib()
    tab = c2.random_table(137, 100, seed=13)
    start, period, ngptotg = 0, None, None
    if split is not None:  # rank `irank` of `numproc`, table covering the global domain (rank offsets apply)
        ngptotg, irank, numproc = split
        s1, e1, size = reflib.get_offsets(irank, numproc, 100, ngptot, ngptotg)
        start, period = B.expand_offsets(100, ngptot, ngptotg, irank, numproc)
        assert (start, period) == (s1 - 1, size)
    ds = c2.DeviceState.from_table(tab, nproma, ngptot, start=start, period=period)
    torch.cuda.synchronize()
    sl = slice(start, start + (100 if period is None else period))  # the columns LOAD_ARRAY hands to EXPAND
    for name in ("PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PSUPSAT"):
        assert np.array_equal(getattr(ds, name).cpu().numpy(), reflib.expand(tab[name][:, sl], nproma, ngptot)), name
    zero = np.zeros_like(tab["PT"])
    pclv = np.stack([tab["PCLV_QL"], tab["PCLV_QI"], zero, zero, zero])
    assert np.array_equal(ds.PCLV.cpu().numpy(), reflib.expand(pclv[:, :, sl], nproma, ngptot))

    # something to validate: NL outputs, against a reference table that differs from them in the 9th digit
    prm = c2.default_params(c2.ceta_from_table(tab))
    ds.nl(prm)
    torch.cuda.synchronize()
    st = ds.download(c2.state_from_table(tab, nproma, ngptot, col0=start))  # (start + g < KLON for a rank slice: the same columns)
    cols = (start + np.arange(min(100, ngptot)) % (100 if period is None else period)) % 100
    rng = np.random.default_rng(3)
    fields = {"PLUDE": st.PLUDE, "PCOVPTOT": st.PCOVPTOT, "PFPLSL": st.PFPLSL, "PFPLSN": st.PFPLSN, "PFHPSL": st.PFHPSL,
              "PFHPSN": st.PFHPSN, "TENDENCY_LOC_A": st.B_LOC[:, 1], "TENDENCY_LOC_Q": st.B_LOC[:, 2],
              "TENDENCY_LOC_T": st.B_LOC[:, 0], "TENDENCY_LOC_CLD": st.B_LOC[:, 3:8]}
    ref_tab = {}
    for name, f in fields.items():  # a KLON-column table whose tiling is the field, perturbed
        per_col = f.transpose(0, 2, 1).reshape(-1, f.shape[-2]).T[:, :ngptot] if f.ndim == 3 else \
            f.transpose(1, 2, 0, 3).reshape(f.shape[1], f.shape[2], -1)[:, :, :ngptot]
        t = np.zeros(per_col.shape[:-1] + (100,))
        t[..., cols] = per_col[..., : len(cols)]
        ref_tab[name] = t * (1.0 + 1e-9 * rng.standard_normal(t.shape))
    rows, text = ds.validate(ref_tab, ngptotg=ngptotg, start=start, period=period)
    lines = text.split("\n")[1:]
    for (label, ndim, stats), line, (name, f) in zip(rows, lines, fields.items()):
        blocked_ref = reflib.expand(ref_tab[name][..., sl], nproma, ngptot)
        want = reflib.validate(label, blocked_ref, f, ngptot, ngptotg)
        # name, nD-option, MinValue, MaxValue, AbsMaxErr: the same characters -- except the SIGN of a zero extreme of a field that holds
        # both zeros (the enthalpy fluxes: -0 at the model top, cloudsc2.F90:732-733, +0 in the padding), which depends on the order
        # of the comparisons in MAXVAL / MAX and is the compiler's choice in the reference too
        unsigned = lambda t: t.replace("-0.0000000000000E+00", " 0.0000000000000E+00")  # noqa: E731
        assert unsigned(line[:88]) == unsigned(want[:88]), (label, line, want)
        for a, b in ((line[88:109], want[88:109]), (line[109:130], want[109:130])):
            assert abs(_fortran_real(a.strip()) - _fortran_real(b.strip())) <= 1e-12 * abs(_fortran_real(b.strip())), (label, a, b)
        assert line[130:] == want[130:], (label, line, want)  # the `!!!!` flag


def test_expand_and_validate_reject_bad_arguments():
    import ctypes as C

    import torch

    t = torch.zeros(137 * 100, dtype=torch.float64, device="cuda:0")
    f = torch.zeros(4 * 137 * 32, dtype=torch.float64, device="cuda:0")
    dp = lambda x: C.cast(x.data_ptr(), C.POINTER(C.c_double))  # noqa: E731
    fld = B.Field(f.data_ptr(), 137 * 32)
    ok = (dp(t), 100, 100, 0, 137, 1, 32, 100, fld, None)
    assert B.lib.cloudsc2_expand_launch(*ok) == 0
    for i, bad in ((2, 101), (2, 0), (3, -1), (6, 0), (7, 0)):
        args = list(ok)
        args[i] = bad
        assert B.lib.cloudsc2_expand_launch(*args) == B.CLOUDSC2_EINVAL, (i, bad)
    small = B.Field(f.data_ptr(), 137 * 32 - 1)
    assert B.lib.cloudsc2_expand_launch(*ok[:8], small, None) == B.CLOUDSC2_EINVAL
    torch.cuda.synchronize()


def test_validator_statistics_match_the_host_restatement(tmp_path):
    """input file -> device state -> NL -> reference file -> validation, at a size with a ragged tail."""
    import torch

    tab = c2.synthetic_table()
    prm0 = c2.default_params(c2.ceta_from_table(tab))
    prm0.nlev = 137
    fileio.write_input_file(str(tmp_path / "input.h5"), tab, prm0)
    tab2, prm = fileio.read_input_file(str(tmp_path / "input.h5"))
    nproma, ngptot = 128, 16300
    ds = c2.DeviceState.from_table(tab2, nproma, ngptot)
    ds.nl(prm)
    torch.cuda.synchronize()
    st = ds.download(c2.state_from_table(tab, nproma, ngptot))
    # reference.h5 written from the first 100 columns (WRITE_REFERENCE), read back, validated: the tiling is periodic and
    # the kernels are deterministic, so every error is exactly zero (option code 1) and no line carries "!!!!"
    ref = fileio.reference_table_from_state(st, 100)
    fileio.write_reference_file(str(tmp_path / "reference.h5"), ref)
    ref = fileio.read_reference_file(str(tmp_path / "reference.h5"))
    rows, text = ds.validate(ref)
    lines = text.split("\n")
    assert lines[0] == B.validate_header() and len(lines) == 11
    names = [r[0] for r in rows]
    assert names == ["PLUDE", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "TENDENCY_LOC%A", "TENDENCY_LOC%Q",
                     "TENDENCY_LOC%T", "TENDENCY_LOC%CLD"]  # print order of cloudsc2_array_state_mod.F90:246-256
    for (name, ndim, s), line in zip(rows, lines[1:]):
        assert s[2] == 0.0 and s[3] == 0.0, (name, s)
        assert line.startswith(" " + name.rjust(20) + f" {ndim}D1") and "!!!!" not in line
    # statistics against numpy (VALIDATE_R2/R3) with a reference that differs
    rng = np.random.default_rng(3)
    ref2 = {k: v * (1.0 + 1e-9 * rng.standard_normal(v.shape)) for k, v in ref.items()}
    rows2, text2 = ds.validate(ref2, ngptotg=2 * ngptot)
    fields = {"PLUDE": st.PLUDE, "PCOVPTOT": st.PCOVPTOT, "PFPLSL": st.PFPLSL, "PFPLSN": st.PFPLSN, "PFHPSL": st.PFHPSL,
              "PFHPSN": st.PFHPSN, "TENDENCY_LOC%A": st.B_LOC[:, 1], "TENDENCY_LOC%Q": st.B_LOC[:, 2],
              "TENDENCY_LOC%T": st.B_LOC[:, 0], "TENDENCY_LOC%CLD": st.B_LOC[:, 3:8]}
    keys = dict(zip(fields, ref2))
    for (name, ndim, s), line in zip(rows2, text2.split("\n")[1:]):
        want = host_validate(fields[name], ref2[keys[name]], nproma, ngptot)
        assert s[0] == want[0] and s[1] == want[1] and s[2] == want[2], (name, s, want)
        assert np.allclose(s[3:], want[3:], rtol=1e-12, atol=0), (name, s, want)
        assert line == B.validate_line(name, ndim, s, 2 * ngptot)
    assert "!!!!" in text2
    # a second validation of the same data gives the same bits (fixed-order reduction)
    rows3, _ = ds.validate(ref2, ngptotg=2 * ngptot)
    for a, b in zip(rows2, rows3):
        assert np.array_equal(a[2], b[2])


def golden_reference_table(tab):
    """What reference.h5 would hold for the synthetic input: the unmodified reference Fortran's outputs on the 100
    synthetic columns (tests/golden/nl_synth100.npz, generated by tests/golden/make_golden.py)."""
    import os

    from tests.util import ROOT

    g = np.load(os.path.join(ROOT, "tests", "golden", "nl_synth100.npz"))
    z = np.zeros_like(g["out_tent"])
    return {"PLUDE": tab["PLUDE"], "PCOVPTOT": g["out_covptot"], "PFPLSL": g["out_fplsl"], "PFPLSN": g["out_fplsn"],
            "PFHPSL": g["out_fhpsl"], "PFHPSN": g["out_fhpsn"], "TENDENCY_LOC_A": z, "TENDENCY_LOC_Q": g["out_tenq"],
            "TENDENCY_LOC_T": g["out_tent"], "TENDENCY_LOC_CLD": np.stack([g["out_tenl"], g["out_teni"], z, z, z])}


def test_gpu_results_pass_the_references_own_validation():
    """VALIDATE's criterion (validate_mod.F90:286: L1 relative error <= 10 eps per variable, else "!!!!") against the
    reference Fortran's outputs, in both math modes."""
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    ref = golden_reference_table(tab)
    try:
        for precise in (False, True):
            c2.set_math_mode(precise)
            ds = c2.DeviceState.from_table(tab, 128, 16384)
            ds.nl(prm)
            rows, text = ds.validate(ref)
            assert "!!!!" not in text, text
            for name, _, s in rows:
                rel = s[3] / s[4] if s[4] > 0 else s[3]
                assert rel <= 10 * np.finfo(np.float64).eps, (precise, name, rel)
    finally:
        c2.set_math_mode(False)


def test_fortran_main_loads_validates_and_writes_reference(tmp_path):
    """dwarf-cloudsc2-nl with the flow of the reference's main (dwarf_cloudsc.F90:79-124): GLOBAL_STATE%LOAD from input.h5,
    CLOUDSC_DRIVER, GLOBAL_STATE%VALIDATE against reference.h5 in the reference's table format, WRITE_REFERENCE."""
    import os

    import torch

    from tests.test_gpu_parity import _run_fortran

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    prm.nlev = 137
    ref = golden_reference_table(tab)
    run = tmp_path / "run"
    run.mkdir()
    fileio.write_input_file(str(run / "input.h5"), tab, prm)
    fileio.write_reference_file(str(run / "reference.h5"), ref)
    nproma, ngptot = 128, 16300
    c2.set_math_mode(False)  # parent and children in the same (default) math mode whatever CLOUDSC2_MATH says
    fast = {"CLOUDSC2_MATH": "fast"}
    out, _ = _run_fortran("dwarf-cloudsc2-nl", 1, ngptot, nproma, cwd=str(run), env=fast)
    lines = [ln for ln in out.split("\n") if ln.strip()]
    i0 = next(i for i, ln in enumerate(lines) if ln == B.validate_header())
    table = lines[i0 + 1:i0 + 11]
    # the same validation on the device through the Python driver
    ds = c2.DeviceState.from_table(tab, nproma, ngptot)
    ds.nl(prm)
    rows, text = ds.validate(ref)
    assert "!!!!" not in out
    for got, want, (name, ndim, s) in zip(table, text.split("\n")[1:], rows):
        assert got[:25] == want[:25], (got, want)                      # name, rank, option code
        g = [float(x) for x in got[25:130].split()]
        w = [float(x) for x in want[25:130].split()]
        assert g[:3] == w[:3], (got, want)  # min, max, max abs error: the same doubles (a zero may differ in sign)
        assert np.allclose(g[3:], w[3:], rtol=1e-9, atol=1e-300), (got, want)  # the sums: sequential vs tree order
    # WRITE_REFERENCE: any NPROMA (the reference insists on 100), equal to the table taken from the Python driver's state
    wr = tmp_path / "write"
    wr.mkdir()
    fileio.write_input_file(str(wr / "input.h5"), tab, prm)
    out, _ = _run_fortran("dwarf-cloudsc2-nl", 1, 300, 64, cwd=str(wr), env={"CLOUDSC2_WRITE_REFERENCE": "1", **fast})
    assert os.path.exists(wr / "reference.h5")
    back = fileio.read_reference_file(str(wr / "reference.h5"))
    st = c2.state_from_table(tab, 64, 300)
    c2.run_state(prm, st, "nl")
    mine = fileio.reference_table_from_state(st, 100)
    for n in fileio.REFERENCE_FIELDS:
        assert np.array_equal(back[n], mine[n]), n
    # and a run validated against the file it has just written reports exact agreement
    out, _ = _run_fortran("dwarf-cloudsc2-nl", 1, 1000, 32, cwd=str(wr), env=fast)
    import re

    lines = [ln for ln in out.split("\n") if re.fullmatch(r"\dD\d", ln[22:25])]
    assert len(lines) == 10 and all(ln[22:25] in ("2D1", "3D1") for ln in lines), out
    torch.cuda.synchronize()


def test_cpp_host_example_prints_the_same_report(tmp_path):
    """examples/cloudsc2_resident.cpp (plain C++ on the C ABI: file -> device tiling -> NL -> device validation) must
    print exactly the report the Python driver produces for the same files."""
    import os
    import subprocess

    from tests.util import ROOT

    exe = os.path.join(ROOT, "examples", "build", "cloudsc2_resident")
    if not os.path.exists(exe):
        pytest.fail(f"{exe} missing: run __graft_entry__.build()")
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    prm.nlev = 137
    ref = golden_reference_table(tab)
    fileio.write_input_file(str(tmp_path / "input.h5"), tab, prm)
    fileio.write_reference_file(str(tmp_path / "reference.h5"), ref)
    nproma, ngptot = 128, 20000
    c2.set_math_mode(False)
    r = subprocess.run([exe, str(ngptot), str(nproma)], cwd=str(tmp_path), capture_output=True, text=True, timeout=300,
                       env={**os.environ, "CLOUDSC2_MATH": "fast"})
    assert r.returncode == 0, r.stdout + r.stderr  # 4 would mean a "!!!!" line
    assert "columns/s" in r.stderr
    tab2, prm2 = fileio.read_input_file(str(tmp_path / "input.h5"))
    ds = c2.DeviceState.from_table(tab2, nproma, ngptot)
    ds.nl(prm2)
    _, text = ds.validate(ref)
    assert r.stdout.rstrip("\n") == text


@pytest.mark.parametrize("nproma,ngptot", [(96, 1000), (32, 1000), (100, 1000), (128, 1000), (1, 100), (1000, 2500), (100, 256)])
def test_resident_state_handle_equals_the_host_pointer_drivers(nproma, ngptot):
    """cloudsc2_state_* (the library-owned resident GLOBAL_STATE the Fortran mains use with CLOUDSC2_RESIDENT=1): expand from
    the KLON-column tables, NL, download == the host-pointer driver on the host-tiled state, bit for bit; the two self-tests
    return the same norms as cloudsc2_tl_taylor_run / cloudsc2_ad_symmetry_run; validation against the state's own outputs
    gives zero error; upload of a host state gives the same results as expand.  The device arrays are blocked by the library
    (128 unless the caller's NPROMA is a multiple of 64): host arrays, the Taylor statistic's blocks and the validator's
    MINVAL / MAXVAL keep the caller's blocking."""
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    want = c2.state_from_table(tab, nproma, ngptot, poison_outputs=3.0)
    c2.run_state(prm, want, "nl")

    rs = c2.ResidentState.from_table(tab, nproma, ngptot)
    assert rs.blocking() == (nproma if nproma % 64 == 0 else 128, nproma)
    ms = rs.nl(prm, repeats=3)
    assert ms > 0.0
    got = rs.download(c2.state_from_table(tab, nproma, ngptot, poison_outputs=3.0))
    for n in ("PA", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "B_LOC"):
        assert np.array_equal(getattr(want, n), getattr(got, n)), n
    # validate a field against a reference table made of its own first KLON columns: zero error, min/max of the field
    cols = got.PFPLSN.transpose(0, 2, 1).reshape(-1, got.nlev + 1)[:100]  # (column, level)
    st = rs.validate(B.F_FULL["PFPLSN"], np.ascontiguousarray(cols.T))
    assert st[2] == 0.0 and st[3] == 0.0 and st[4] > 0.0
    act = got.PFPLSN.transpose(0, 2, 1).reshape(-1, got.nlev + 1)[:ngptot]
    if got.nblocks * nproma > ngptot:  # MINVAL / MAXVAL see the zero padding (FIELD_INIT) of the CALLER's last block, if it has any
        assert st[0] == min(act.min(), 0.0) and st[1] == max(act.max(), 0.0)
    else:
        assert st[0] == act.min() and st[1] == act.max()
    # the same on an all-positive field (PT): the caller's padded tail is the only zero, and it counts even where the device's own
    # blocking has no padding at all (caller NPROMA 100 pads 256 columns to 300, the device holds 2 x 128)
    tcols = want.PT.transpose(0, 2, 1).reshape(-1, want.nlev)[:100]
    stt = rs.validate(B.F_FULL["PT"], np.ascontiguousarray(tcols.T))
    tact = want.PT.transpose(0, 2, 1).reshape(-1, want.nlev)[:ngptot]
    assert stt[2] == 0.0 and tact.min() > 0.0
    assert stt[0] == (0.0 if got.nblocks * nproma > ngptot else tact.min()) and stt[1] == tact.max()

    # the self-tests on the resident state vs. on host arrays
    prm_tl = c2.default_params(c2.ceta_from_table(tab), lregcl=False)
    zn_h, ok_h, it_h, _ = c2.run_state(prm_tl, c2.state_from_table(tab, nproma, ngptot), "tl")
    zn_r, ok_r, it_r, _ = rs.tl_taylor(prm_tl)
    assert np.array_equal(zn_h, zn_r) and ok_h == ok_r and it_h == it_r
    prm_ad = c2.default_params(c2.ceta_from_table(tab), lregcl=True)
    za_h, oka_h, _ = c2.run_state(prm_ad, c2.state_from_table(tab, nproma, ngptot), "ad")
    za_r, oka_r, _ = rs.ad_symmetry(prm_ad)
    assert za_h == za_r and oka_h and oka_r

    rs2 = c2.ResidentState(nproma, tab["PT"].shape[0], ngptot)
    rs2.ptsphy = rs.ptsphy
    rs2.upload(c2.state_from_table(tab, nproma, ngptot))
    rs2.nl(prm)
    got2 = rs2.download(c2.state_from_table(tab, nproma, ngptot, poison_outputs=3.0))
    assert np.array_equal(got2.B_LOC, want.B_LOC) and np.array_equal(got2.PFHPSN, want.PFHPSN)
    with pytest.raises(c2.Cloudsc2Error):
        rs.validate(99, np.zeros((138, 100)))


def test_state_allocation_is_judged_by_the_nl_sweep():
    """cloudsc2_device_malloc_state: the candidates of a state's allocation are timed with the NL kernel itself on a state laid out
    in each of them; the Python mirror's DeviceState lives in such an allocation, in the library's own array order, and computes the
    same results as before (checked against the reference by the parity tests; here: against the flat host-side layout)."""
    import ctypes as C

    nproma, nlev, ngptot = 128, 137, 60000
    nb = (ngptot + nproma - 1) // nproma
    need = B.DeviceArena.size_of([(nb, nlev, nproma)] * 11 + [(nb, nlev + 1, nproma)] * 5 + [(nb, 8, nlev, nproma)] * 2 + [(nb, 5, nlev, nproma)],
                                 B.REAL_BYTES)
    p = C.c_void_p()
    assert B.lib.cloudsc2_device_malloc_state(C.byref(p), need - 4096, nproma, nlev, ngptot) == B.CLOUDSC2_EINVAL  # too small for the state
    buf = B.DeviceBuffer(need + (64 << 20), (nproma, nlev, ngptot))
    info = B.device_malloc_info()
    assert info["candidates"] >= 4 and info["probe_ms_best"] > 0.0
    del buf
    tab = c2.synthetic_table()
    ds = c2.DeviceState.from_table(tab, nproma, ngptot, "cuda:0")
    ptrs = [getattr(ds, n).data_ptr() for n in ds.ORDER]
    assert ptrs == sorted(ptrs) and ptrs[0] == ds.arena.buf.ptr  # the library's order, from the start of the arena
    assert ds.arena.info["candidates"] >= 4


def test_device_allocator_places_and_frees():
    """cloudsc2_device_malloc / _free / _malloc_info (include/cloudsc2_hip.h): a large request is placed (several candidates probed),
    a small one is not, a foreign pointer is refused, and the memory is ordinary device memory."""
    import ctypes as C

    import torch

    big = B.DeviceBuffer(3 << 30)
    info = B.device_malloc_info()
    assert info["candidates"] >= 4 and 0.0 < info["probe_ms_best"] <= info["probe_ms_worst"] and info["probe_ms_median"] <= info["probe_ms_worst"]
    assert B.device_probe(big.ptr, big.nbytes, 0, 3) > 0.0 and B.device_probe(big.ptr, big.nbytes, 1, 3) > 0.0  # the diagnostic streams
    t = torch.as_tensor(big, device="cuda:0")
    t[:1024].fill_(7)
    assert int(t[:1024].sum().item()) == 7 * 1024
    small = B.DeviceBuffer(1 << 20)
    assert B.device_malloc_info()["candidates"] == 1
    del t, big, small
    rc = B.lib.cloudsc2_device_free(C.c_void_p(0x1000))
    assert rc == B.CLOUDSC2_EINVAL
    # the composed form (requests above 12 GiB by default; forced here for a small one)
    import os
    import subprocess
    import sys

    from tests.util import ROOT

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\nfrom dwarf_p_cloudsc2_tl_ad_amd import binding as B\n"
            "b = B.DeviceBuffer(5 << 30); i = B.device_malloc_info(); t = torch.as_tensor(b, device='cuda:0')\n"
            "t[-4096:].fill_(3); torch.cuda.synchronize(); print('OK', i['candidates'], int(t[-4096:].sum().item()))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "CLOUDSC2_PLACE_MODE": "chunks"}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split()[1]) >= 11 and int(r.stdout.split()[2]) == 3 * 4096
    # the search's transient footprint is bounded where the host model says so, and nothing is searched on a device that other
    # ranks allocate on at the same time (hipMemGetInfo and hipMalloc are not coordinated between processes)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dwarf_p_cloudsc2_tl_ad_amd import binding as B\n"
            "b = B.DeviceBuffer(3 << 30); print('OK', B.device_malloc_info()['candidates'])\n" % ROOT)
    for env, want in (({"CLOUDSC2_PLACE_MAX_GB": "7"}, 2), ({"CLOUDSC2_PLACE_SHARED": "1"}, 1), ({"LOCAL_WORLD_SIZE": "64"}, 1),
                      ({"CLOUDSC2_PLACE_MAX_SHARE": "0.001"}, 1)):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.split()[:1] == ["OK"], r.stdout + r.stderr
        # (the second batch of candidates, taken when the first shows no class difference, shares the budget: at most `want`)
        assert 1 <= int(r.stdout.split()[1]) <= want, (env, r.stdout)


def test_host_array_drivers_do_not_search_for_a_place():
    """cloudsc2_nl_run / cloudsc2_tl_taylor_run / cloudsc2_ad_symmetry_run are PCIe-bound, so their workspace is
    ONE plain hipMalloc (no candidates, no transient share of the host model's HBM) unless CLOUDSC2_PLACE=1 asks; the resident
    state keeps the search.  Fresh processes (the allocator reads its switches once); 20 000 columns = 0.8 GB, above the 256 MiB
    below which nothing is searched anyway."""
    import subprocess
    import sys

    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import dwarf_p_cloudsc2_tl_ad_amd as c2\n"
        "from dwarf_p_cloudsc2_tl_ad_amd import binding as B\n"
        "tab = c2.synthetic_table(); ceta = c2.ceta_from_table(tab); out = {}\n"
        "st = c2.state_from_table(tab, 128, 20000)\n"
        "c2.run_state(c2.default_params(ceta), st, 'nl'); out['nl'] = B.device_malloc_counts()\n"
        "c2.run_state(c2.default_params(ceta, lregcl=False), c2.state_from_table(tab, 128, 20000), 'tl'); out['tl'] = B.device_malloc_counts()\n"
        "c2.run_state(c2.default_params(ceta, lregcl=True), c2.state_from_table(tab, 128, 20000), 'ad'); out['ad'] = B.device_malloc_counts()\n"
        "B.lib.cloudsc2_release_workspace()\n"
        "rs = c2.ResidentState.from_table(tab, 128, 20000); out['resident'] = B.device_malloc_counts()\n"
        "print(json.dumps(out))\n" % ROOT)

    def run(env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("CLOUDSC2_PLACE")}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env={**e, **env})
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]), r.stderr

    d, err = run({"CLOUDSC2_PLACE_VERBOSE": "1"})
    assert d["nl"][0] == 0 and d["nl"][1] >= 1, d           # one plain allocation, nothing searched
    assert d["tl"][0] == 0 and d["ad"][0] == 0, d           # (the TL / AD workspaces are larger: re-allocated, still plain)
    assert d["resident"][0] == 1, d                          # cloudsc2_state_create: searched
    assert "host-array driver workspace: searched only with CLOUDSC2_PLACE=1" in err
    d1, _ = run({"CLOUDSC2_PLACE": "1"})
    assert d1["nl"][0] == 1, d1                              # asked for explicitly: searched
    d0, _ = run({"CLOUDSC2_PLACE": "0"})
    assert d0["resident"][0] == 0 and d0["nl"][0] == 0, d0  # every search off
