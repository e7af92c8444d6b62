"""usage: python tools/debug/evap_diff.py  -- where do the NL outputs of the evaporation variant differ from the checker's?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from tests.test_gpu_parity import checker, ref_nl_state  # noqa: E402
from tests.util import c2, set_lib_params  # noqa: E402

tab = c2.random_table(137, 64, seed=11)
prm = c2.default_params(c2.ceta_from_table(tab), levapls2=True)
st = c2.state_from_table(tab, 64, 128)
chk = checker()
set_lib_params(chk, prm)
want = ref_nl_state(chk, st, prm)
got = st.copy()
c2.run_state(prm, got, "nl")
for name in ("PCOVPTOT", "PFPLSL", "PFPLSN", "PA"):
    w, g = getattr(want, name), getattr(got, name)
    d = np.abs(w - g)
    idx = np.argwhere(d > 1e-9 * max(1e-300, np.abs(w).max()))
    print(name, "n differing", len(idx), "max", d.max())
    for ib, jk, jl in [i for i in idx if i[1] + 1 < w.shape[1]][:6]:
        print("  block", ib, "level", jk, "col", jl, "want", w[ib, jk, jl], "got", g[ib, jk, jl],
              "| PA", want.PA[ib, jk, jl], "fluxes in (want)", want.PFPLSL[ib, jk, jl], want.PFPLSN[ib, jk, jl],
              "out", want.PFPLSL[ib, jk + 1, jl], want.PFPLSN[ib, jk + 1, jl], "got out", got.PFPLSL[ib, jk + 1, jl], got.PFPLSN[ib, jk + 1, jl],
              "covptot above", want.PCOVPTOT[ib, jk - 1, jl], got.PCOVPTOT[ib, jk - 1, jl])
