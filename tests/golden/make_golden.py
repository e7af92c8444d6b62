"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference Fortran (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

The fixtures hold data only: a checksum of the (deterministic, regenerated) inputs and the reference's outputs.
  nl_synth100.npz      SATUR + CLOUDSC2 on the 100-column synthetic atmosphere (the shape of reference.h5)
  tlad_synth24_r{0,1}  CLOUDSC2TL outputs for dx = 0.01 x and CLOUDSC2AD input adjoints for y = TL dx, LREGCL off/on
  evap_rand24.npz      the same three kernels with LEVAPLS2=.true. (the block that is dead in the shipped configs)
  drivers.json         what the reference's own TL and AD test drivers print (Taylor ratios, verdicts, max AD error)
  table_numpy.npz      the fields of the synthetic table as numpy evaluates the recipe (rounds 1-4), where they differ from the library's
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.util import c2, make_params, refcall, set_lib_params  # noqa: E402


def input_digest(inp: dict) -> str:
    h = hashlib.sha256()
    for n in refcall.IN16:
        h.update(np.ascontiguousarray(inp[n]).tobytes())
    return h.hexdigest()


def table_inputs(ref, tab, ncol):
    """One NPROMA=ncol block of kernel inputs (KLEV, KLON) from a table, with the reference's SATUR for PQS."""
    st = c2.state_from_table(tab, ncol, ncol)
    qs = ref.satur(np.ascontiguousarray(st.PAP[0]), np.ascontiguousarray(st.PT[0]))
    return st, refcall.block_inputs(st, 0, qs)


def capture_stdout(fn):
    sys.stdout.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            fn()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode(errors="replace")


def tl_ad(ref, ptsphy, inp, ldrain1d=False):
    dinp = {n: np.ascontiguousarray(a * 0.01) for n, a in inp.items()}
    out5, dout = ref.cloudsc2tl(ptsphy, inp, dinp, ldrain1d=ldrain1d)
    x = refcall.new_inputs(*inp["pap"].shape)
    y = {n: a.copy() for n, a in dout.items()}
    ref.cloudsc2ad(ptsphy, inp, x, y, ldrain1d=ldrain1d)
    return out5, dout, x


def numpy_flavoured_table(nlev=137, ncol=100):
    """The synthetic atmosphere as rounds 1-4 generated it: the same recipe as cloudsc2_synthetic_table evaluated by numpy, whose
    exp / power differ from glibc's in the last place (PQ up to 9e-15 relative).  Kept as a FIXTURE (table_numpy.npz), not as code the
    product uses: on this table the reference's own Taylor test FAILS at NPROMA 32 (one wiggle in the round-off arm), on the
    library's table it passes -- the failing verdict is pinned on the data that produces it."""
    rd, rv, rtt = 287.0597, 461.5250, 273.16
    r2es, r3les, r4les = 611.21 * rd / rv, 17.502, 32.19
    ig = np.arange(ncol, dtype=np.int64)
    h1, h2, h3 = ((37 * ig) % 100) / 100.0, ((61 * ig + 13) % 100) / 100.0, ((89 * ig + 7) % 100) / 100.0
    k = np.arange(nlev + 1, dtype=np.float64)
    ps = 101325.0
    paph_1d = 1.0 + (ps - 1.0) * (k / nlev) ** 2.2
    pap_1d = 0.5 * (paph_1d[:-1] + paph_1d[1:])
    paph, pap = np.repeat(paph_1d[:, None], ncol, axis=1), np.repeat(pap_1d[:, None], ncol, axis=1)
    eta = pap / ps
    t = np.maximum(205.0 + 10.0 * h2[None, :], (255.0 + 45.0 * h1[None, :]) * eta**0.19)
    rh = 0.35 + (0.72 + 0.1 * h3[None, :]) * np.exp(-(((eta - 0.3 - 0.5 * h2[None, :]) / 0.18) ** 2))
    e_liq = r2es * np.exp(r3les * (t - rtt) / (t - r4les))
    q = rh * np.minimum(0.5, e_liq / pap)
    moist = rh > 0.8
    ql = 1e-7 * eta + np.where(moist, 2e-5 * h1[None, :] * eta, 0.0)
    qi = 1e-7 * (1.0 - eta) + np.where(moist, 1e-5 * (1.0 - h1[None, :]), 0.0)
    conv = (h3[None, :] > 0.6) & (eta > 0.35) & (eta < 0.9)
    zeros = np.zeros((nlev, ncol))
    return {"PT": t, "PQ": q, "PAP": pap, "PAPH": paph, "PLU": np.where(conv, 3e-4 * h3[None, :], 0.0),
            "PLUDE": np.where(conv & (eta < 0.5), 1e-6 * h3[None, :], 0.0), "PMFU": np.where(conv, 0.05 * h3[None, :], 0.0),
            "PMFD": np.where(conv, -0.01 * h3[None, :], 0.0), "PA": zeros.copy(), "PCLV_QL": ql, "PCLV_QI": qi, "PSUPSAT": zeros.copy(),
            "TENDENCY_CML_T": np.repeat((1e-5 * (h1 - 0.5))[None, :], nlev, axis=0),
            "TENDENCY_CML_Q": np.repeat((1e-9 * (h2 - 0.5))[None, :], nlev, axis=0), "TENDENCY_CML_QL": zeros.copy(),
            "TENDENCY_CML_QI": zeros.copy(), "PTSPHY": 3600.0}


def main():
    ref = refcall.RefLib()
    tab = c2.synthetic_table()
    meta = {}

    # ---- NL, 100 columns ----
    prm = make_params(tab)
    set_lib_params(ref, prm)
    st, inp = table_inputs(ref, tab, 100)
    out = ref.cloudsc2(st.ptsphy, inp)
    np.savez_compressed(os.path.join(HERE, "nl_synth100.npz"), qsat=inp["qsat"], **{"out_" + n: a for n, a in out.items()})
    meta["nl_synth100"] = {"digest": input_digest(inp), "ncol": 100, "flags": {}}

    # ---- TL / AD, 24 columns, both regularisation settings ----
    tab24 = {k: (v[:, :24] if isinstance(v, np.ndarray) else v) for k, v in tab.items()}
    for lreg in (0, 1):
        prm = make_params(tab24, lregcl=bool(lreg))
        set_lib_params(ref, prm)
        st, inp = table_inputs(ref, tab24, 24)
        out5, dout, x = tl_ad(ref, st.ptsphy, inp)
        np.savez_compressed(os.path.join(HERE, f"tlad_synth24_r{lreg}.npz"), qsat=inp["qsat"],
                            **{"tl_" + n: a for n, a in dout.items()}, **{"ad_" + n: a for n, a in x.items()},
                            **{"traj_" + n: a for n, a in out5.items()})
        meta[f"tlad_synth24_r{lreg}"] = {"digest": input_digest(inp), "ncol": 24, "flags": {"lregcl": bool(lreg)}}

    # ---- evaporation branch on (LEVAPLS2), random atmosphere ----
    rt = c2.random_table(137, 24, seed=7)
    prm = make_params(rt, levapls2=True, lregcl=True)
    set_lib_params(ref, prm)
    st, inp = table_inputs(ref, rt, 24)
    out = ref.cloudsc2(st.ptsphy, inp)
    out5, dout, x = tl_ad(ref, st.ptsphy, inp)
    assert np.any(out["covptot"] != 0.0), "evaporation branch not exercised"
    np.savez_compressed(os.path.join(HERE, "evap_rand24.npz"), qsat=inp["qsat"], **{"out_" + n: a for n, a in out.items()},
                        **{"tl_" + n: a for n, a in dout.items()}, **{"ad_" + n: a for n, a in x.items()})
    meta["evap_rand24"] = {"digest": input_digest(inp), "ncol": 24, "flags": {"levapls2": True, "lregcl": True}, "seed": 7}

    # ---- the reference's own test drivers ----
    drivers = {}
    prm = make_params(tab, lregcl=False)
    set_lib_params(ref, prm)
    # (32, 800) and (128, 3200): the FULL set of distinct blocks of the 100-periodic state at that blocking (lcm(NPROMA, 100) columns),
    # i.e. the statistic of any larger run -- 160 000 columns included -- since ZNORMG is a MAX over blocks (cloudsc_driver_tl_mod.F90:249).
    # At NPROMA 32 the reference's own verdict is FAILLED (one wiggle in the round-off arm of the V-shape test): pinned as it is.
    for nproma, ngptot in ((32, 100), (1, 100), (32, 800), (128, 3200)):
        st = c2.state_from_table(tab, nproma, ngptot)
        txt = capture_stdout(lambda: ref.driver(1, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))  # noqa: B023
        ratios = [float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", txt, flags=re.M)][:10]
        verdict = re.search(r"TEST (PASSED|FAILLED).*", txt).group(0).strip()
        drivers[f"tl_nproma{nproma}_ngptot{ngptot}"] = {"znormg": ratios, "verdict": verdict}
    # the same driver on the numpy-flavoured table (fixture table_numpy.npz): the reference's FAILING verdict, pinned with the data
    ntab = numpy_flavoured_table()
    lib_tab = c2.synthetic_table()
    differing = [n for n in ntab if isinstance(ntab[n], np.ndarray) and not np.array_equal(ntab[n], lib_tab[n])]
    np.savez_compressed(os.path.join(HERE, "table_numpy.npz"), **{n: ntab[n] for n in differing})
    meta["table_numpy"] = {"fields_that_differ_from_cloudsc2_synthetic_table": differing,
                           "max_relative_difference": {n: float(np.max(np.abs(ntab[n] - lib_tab[n]) / np.abs(lib_tab[n]))) for n in differing}}
    prm = make_params(ntab, lregcl=False)
    set_lib_params(ref, prm)
    for nproma, ngptot in ((32, 800), (128, 3200)):
        st = c2.state_from_table(ntab, nproma, ngptot)
        txt = capture_stdout(lambda: ref.driver(1, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))  # noqa: B023
        ratios = [float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", txt, flags=re.M)][:10]
        verdict = re.search(r"TEST (PASSED|FAILLED).*", txt).group(0).strip()
        drivers[f"tl_nproma{nproma}_ngptot{ngptot}_numpy_table"] = {"znormg": ratios, "verdict": verdict}
    prm = make_params(tab, lregcl=True)
    set_lib_params(ref, prm)
    for nproma, ngptot in ((100, 100), (64, 1000)):
        st = c2.state_from_table(tab, nproma, ngptot)
        txt = capture_stdout(lambda: ref.driver(2, 1, nproma, st.nlev, ngptot, st.ptsphy, st.driver_arrays()))  # noqa: B023
        err = float(re.search(r"maximum error is\s+([0-9.Ee+-]+)", txt).group(1))
        verdict = "OK" if "TEST OK" in txt else "FAILED"
        drivers[f"ad_nproma{nproma}_ngptot{ngptot}"] = {"znormg": err, "verdict": verdict}
    json.dump({"fixtures": meta, "drivers": drivers}, open(os.path.join(HERE, "drivers.json"), "w"), indent=1)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
    print(json.dumps(drivers, indent=1))


if __name__ == "__main__":
    main()
