"""One-off randomised parity sweep on the GPU (not collected by pytest): python tests/fuzz_parity.py [N [SEED]].
Random atmospheres, NLEV, NPROMA, ragged NGPTOT and switches; NL through the driver-level C ABI and TL/AD at kernel
level against the CPU checker, with the tolerances of tests/test_gpu_parity.py."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import NL_TOL, TLAD_TOL, _device_tl_ad, assert_outputs_close, checker, ref_nl_state  # noqa: E402
from tests.util import c2, relerr, set_lib_params  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
worst = {"nl": 0.0, "tl": 0.0, "ad": 0.0}
for it in range(n):
    nlev = int(rng.choice([137, 137, 91, 60, 30, 200]))
    ncol = int(rng.integers(20, 90))
    nproma = int(rng.choice([1, 7, 16, 33, 64, 100, 128, 192]))
    ngptot = int(rng.integers(max(2, nproma // 2), 3 * nproma + 40))
    flags = dict(lregcl=bool(rng.integers(2)), levapls2=bool(rng.integers(2)), ldrain1d=bool(rng.integers(4) == 0))
    tab = c2.random_table(nlev, ncol, seed=int(rng.integers(1 << 30)))
    prm = c2.default_params(c2.ceta_from_table(tab), **flags)
    st = c2.state_from_table(tab, nproma, ngptot)
    chk = checker()
    set_lib_params(chk, prm)
    want = ref_nl_state(chk, st, prm)
    got = st.copy()
    c2.run_state(prm, got, "nl")
    # random atmospheres: ten times the suite's NL tolerance -- the fast arithmetic's few ulp are amplified where the cloud cover
    # saturates (seen: 2.2e-12 in PA at cover 0.987 with LREGCL, 3.6e-14 in precise arithmetic: tools/fuzz_case.py 38 777); a
    # flipped branch would show at 1e-6 or more
    assert_outputs_close(want, got, 10 * NL_TOL)
    worst["nl"] = max(worst["nl"], max(relerr(r, got.outputs()[k]) for k, r in want.outputs().items()))
    r = _device_tl_ad(tab, nproma, ngptot, flags)
    act = lambda a: np.concatenate([a[ibl][:, : min(nproma, ngptot - ibl * nproma)] for ibl in range(st.nblocks)], axis=1)  # noqa: E731
    e_tl = max(relerr(act(r["tl_ref"][k]), act(r["tl_dev"][k])) for k in r["tl_ref"])
    e_ad = 0.0
    for k in r["x_ref"]:
        ref_inc = act(r["x_ref"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_ref"][k])
        got_inc = act(r["x_dev"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_dev"][k])
        e_ad = max(e_ad, float(np.abs(got_inc - ref_inc).max() / max(np.abs(act(r["x_ref"][k])).max(), 1e-300)))
    assert e_tl <= TLAD_TOL and e_ad <= TLAD_TOL, (it, e_tl, e_ad)
    worst["tl"], worst["ad"] = max(worst["tl"], e_tl), max(worst["ad"], e_ad)
    print(f"case {it:2d}: nlev {nlev:3d} nproma {nproma:3d} ngptot {ngptot:4d} {flags}  tl {e_tl:.1e} ad {e_ad:.1e}", flush=True)
print("FUZZ PASSED", n, "cases; worst nl", f"{worst['nl']:.2e}", "tl", f"{worst['tl']:.2e}", "ad", f"{worst['ad']:.2e}")
