"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference Fortran (oracle/_ref, built by
oracle/Makefile from /root/reference).  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden.py

The fixtures hold data only: a checksum of the (deterministic, regenerated) inputs and the reference's outputs.
  nl_synth100.npz      SATUR + CLOUDSC2 on the 100-column synthetic atmosphere (the shape of reference.h5)
  tlad_synth24_r{0,1}  CLOUDSC2TL outputs for dx = 0.01 x and CLOUDSC2AD input adjoints for y = TL dx, LREGCL off/on
  evap_rand24.npz      the same three kernels with LEVAPLS2=.true. (the block that is dead in the shipped configs)
  drivers.json         what the reference's own TL and AD test drivers print (Taylor ratios, verdicts, max AD error)
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
    for nproma in (32, 1):
        st = c2.state_from_table(tab, nproma, 100)
        txt = capture_stdout(lambda: ref.driver(1, 1, nproma, st.nlev, 100, st.ptsphy, st.driver_arrays()))  # noqa: B023
        ratios = [float(m.group(1)) for m in re.finditer(r"^\s*\d+\s+([0-9.Ee+-]+)\s*$", txt, flags=re.M)][:10]
        verdict = re.search(r"TEST (PASSED|FAILLED).*", txt).group(0).strip()
        drivers[f"tl_nproma{nproma}_ngptot100"] = {"znormg": ratios, "verdict": verdict}
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
