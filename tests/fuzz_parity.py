"""The long randomised parity sweep on the GPU (by hand; a fixed slice of it is tests/test_gpu_fuzz.py): python tests/fuzz_parity.py
[N [SEED]].  Random atmospheres, NLEV, NPROMA, ragged NGPTOT and switches; NL through the driver-level C ABI in fast and precise
arithmetic, TL / AD at kernel level, against the CPU checker."""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.fuzz_cases import cases, run_case  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
worst = {}
for c in cases(seed, n):
    err = run_case(c, nl_tol_fast=1e-11, nl_tol_precise=1e-12, tlad_tol=1e-11)
    for k, v in err.items():
        worst[k] = max(worst.get(k, 0.0), v)
    print(f"case {c['index']:2d}: nlev {c['nlev']:3d} nproma {c['nproma']:3d} ngptot {c['ngptot']:4d} {c['flags']}  "
          + " ".join(f"{k} {v:.1e}" for k, v in err.items()), flush=True)
print("FUZZ PASSED", n, "cases; worst", " ".join(f"{k} {v:.2e}" for k, v in worst.items()))
