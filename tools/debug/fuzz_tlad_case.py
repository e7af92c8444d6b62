"""usage: [CLOUDSC2_MATH=precise] python tools/debug/fuzz_tlad_case.py CASE SEED  -- one case of tests/fuzz_parity.py again, TL and AD field by field:
the largest |difference| from the checker relative to the field's largest value."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from tests.fuzz_cases import cases  # noqa: E402
from tests.test_gpu_parity import _device_tl_ad  # noqa: E402
from tests.util import c2  # noqa: E402

case, seed = int(sys.argv[1]), int(sys.argv[2])
c = cases(seed, case + 1)[case]
print(c, "math default:", os.environ.get("CLOUDSC2_MATH", "fast"))
tab = c2.random_table(c["nlev"], c["ncol"], seed=c["table_seed"])
nproma, ngptot = c["nproma"], c["ngptot"]
r = _device_tl_ad(tab, nproma, ngptot, c["flags"])
nb = (ngptot + nproma - 1) // nproma
act = lambda a: np.concatenate([a[ibl][:, : min(nproma, ngptot - ibl * nproma)] for ibl in range(nb)], axis=1)  # noqa: E731
for k in r["tl_ref"]:
    a, b = act(r["tl_ref"][k]), act(r["tl_dev"][k])
    i = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
    print(f"TL {k:8s} err {np.abs(a - b).max() / max(np.abs(a).max(), 1e-300):.2e}  at level {i[0]} col {i[1]}: ref {a[i]:.6e} (field max {np.abs(a).max():.3e})")
for k in r["x_ref"]:
    ref_inc = act(r["x_ref"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_ref"][k])
    got_inc = act(r["x_dev"][k]) - act(r["x0"][k]) if k != "supsat" else act(r["x_dev"][k])
    d = np.abs(got_inc - ref_inc)
    i = np.unravel_index(np.argmax(d), d.shape)
    print(f"AD {k:8s} err {d.max() / max(np.abs(act(r['x_ref'][k])).max(), 1e-300):.2e}  at level {i[0]} col {i[1]}: ref increment {ref_inc[i]:.6e} (field max {np.abs(act(r['x_ref'][k])).max():.3e})")
