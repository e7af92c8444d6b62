"""`python bench.py --gpus N` must start N ranks by itself (--gpus was parsed and ignored).  Rehearsed on the CPU
as far as a GPU-less container allows: spawn, rendezvous (gloo), the reference's column split, the verdict reductions, ONE
JSON line from rank 0 -- `--rendezvous-only` skips the device work and nothing else."""
from __future__ import annotations

import json
import os
import subprocess
import sys

from tests.util import ROOT


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *map(str, args)], capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=ROOT)


def test_gpus_flag_starts_that_many_ranks():
    r = _run(["--gpus", 3, "--rendezvous-only", "--ngptot", 1000])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3
    assert d["first_column_per_rank"] == [0.0, 1000.0, 2000.0] and d["columns_per_rank"] == 1000
    # element-wise MAX over the ranks of what each rank contributed (rank r: 1 + 10^-(k+1) (r+1), 5 + r)
    assert abs(d["verdicts"]["tl_znormg"][0] - 1.3) < 1e-12 and d["verdicts"]["ad_znormg"] == 7.0


def test_a_launcher_with_another_world_size_is_refused():
    r = _run(["--gpus", 4, "--rendezvous-only"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_without_a_gpu_the_bench_fails_loudly():
    """No CPU fallback: on a box without a HIP device the measurement refuses to run (this test is skipped on GPU boxes)."""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _run(["--steps", 1, "--warmup", 0, "--no-cpu-baseline", "--no-companions"])
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
