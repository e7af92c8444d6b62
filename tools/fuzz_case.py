#!/usr/bin/env python3
"""One case of tests/fuzz_parity.py again (same random sequence), NL only, in fast and in precise arithmetic: per output field the
largest |difference| relative to the field's largest value, and where it is.  usage: python tools/fuzz_case.py CASE [SEED]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_parity import checker, ref_nl_state  # noqa: E402
from tests.util import c2, set_lib_params  # noqa: E402

from tests.fuzz_cases import cases  # noqa: E402

case = int(sys.argv[1])
c = cases(int(sys.argv[2]) if len(sys.argv) > 2 else 2026, case + 1)[case]
nlev, ncol, nproma, ngptot, flags, seed = c["nlev"], c["ncol"], c["nproma"], c["ngptot"], c["flags"], c["table_seed"]
print(f"case {case}: nlev {nlev} ncol {ncol} nproma {nproma} ngptot {ngptot} {flags} table seed {seed}")
tab = c2.random_table(nlev, ncol, seed=seed)
prm = c2.default_params(c2.ceta_from_table(tab), **flags)
st = c2.state_from_table(tab, nproma, ngptot)
chk = checker()
set_lib_params(chk, prm)
want = ref_nl_state(chk, st, prm)
for mode, name in ((1, "fast"), (2, "precise")):
    prm.math_mode = mode
    got = st.copy()
    c2.run_state(prm, got, "nl")
    print(f"-- {name} arithmetic")
    for n, r in want.outputs().items():
        g = got.outputs()[n]
        d = np.abs(g - r)
        i = np.unravel_index(int(np.argmax(d)), d.shape)
        print(f"   {n:10s} max|diff|/max|ref| {d.max() / max(np.abs(r).max(), 1e-300):.2e} at {i}: reference {r[i]:.17g} here {g[i]:.17g}")
