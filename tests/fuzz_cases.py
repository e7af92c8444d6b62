"""The random cases of the parity fuzz sweep: atmospheres, NLEV, NPROMA, ragged NGPTOT and switches drawn from one seeded
sequence, shared by tests/test_gpu_fuzz.py (a fixed slice, collected with -m gpu), tests/fuzz_parity.py (the long sweep, by hand)
and tools/fuzz_case.py (one case again, field by field)."""
from __future__ import annotations

import numpy as np


def cases(seed: int, n: int) -> list[dict]:
    """The first n cases of the sequence `seed` (case i is the same whatever n is)."""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n):
        nlev = int(rng.choice([137, 137, 91, 60, 30, 200]))
        ncol = int(rng.integers(20, 90))
        nproma = int(rng.choice([1, 7, 16, 33, 64, 100, 128, 192]))
        ngptot = int(rng.integers(max(2, nproma // 2), 3 * nproma + 40))
        flags = dict(lregcl=bool(rng.integers(2)), levapls2=bool(rng.integers(2)), ldrain1d=bool(rng.integers(4) == 0))
        out.append(dict(index=it, nlev=nlev, ncol=ncol, nproma=nproma, ngptot=ngptot, flags=flags, table_seed=int(rng.integers(1 << 30))))
    return out


def run_case(case: dict, nl_tol_fast: float, nl_tol_precise: float, tlad_tol: float) -> dict:
    """NL through the driver-level C ABI in both arithmetic modes, TL / AD at kernel level, against the CPU checker (the reference
    Fortran when oracle/_ref travelled).  Returns the worst relative errors; raises on a tolerance."""
    from tests.test_gpu_parity import _device_tl_ad, assert_outputs_close, checker, ref_nl_state
    from tests.util import c2, relerr, set_lib_params

    nproma, ngptot, flags = case["nproma"], case["ngptot"], case["flags"]
    tab = c2.random_table(case["nlev"], case["ncol"], seed=case["table_seed"])
    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, nproma, ngptot)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    err = {}
    for mode, name, tol in ((1, "nl_fast", nl_tol_fast), (2, "nl_precise", nl_tol_precise)):
        prm.math_mode = mode
        got = st.copy()
        c2.run_state(prm, got, "nl")
        assert_outputs_close(want, got, tol)
        err[name] = max(relerr(r, got.outputs()[k]) for k, r in want.outputs().items())
    r = _device_tl_ad(tab, nproma, ngptot, flags)
    act = lambda a: np.concatenate([a[ibl][:, : min(nproma, ngptot - ibl * nproma)] for ibl in range(st.nblocks)], axis=1)  # noqa: E731
    err["tl"] = max(relerr(act(r["tl_ref"][k]), act(r["tl_dev"][k])) for k in r["tl_ref"])
    err["ad"] = 0.0
    for k in r["x_ref"]:
        ref_inc = act(r["x_ref"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_ref"][k])
        got_inc = act(r["x_dev"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_dev"][k])
        err["ad"] = max(err["ad"], float(np.abs(got_inc - ref_inc).max() / max(np.abs(act(r["x_ref"][k])).max(), 1e-300)))
    assert err["tl"] <= tlad_tol and err["ad"] <= tlad_tol, (case, err)
    return err
