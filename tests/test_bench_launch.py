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


def test_a_rank_that_dies_at_start_fails_the_launch_within_seconds():
    """One rank exits 3 before the rendezvous: the parent stops the others (they would sit in the rendezvous until its
    timeout), exits non-zero, names the rank and relays its stderr -- nothing is restarted."""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", 3, "--rendezvous-only", "--ngptot", 1000], env={"CLOUDSC2_BENCH_FAIL_RANK": "1"}, timeout=120)
    dt = time.monotonic() - t0
    assert r.returncode != 0, r.stdout
    assert dt < 30.0, dt
    assert "(1, 3)" in r.stderr and "told to fail at start" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()]  # no line from a failed launch


def test_the_launch_has_an_overall_deadline():
    """All ranks healthy but slower than the deadline: the parent kills them and fails instead of hanging."""
    r = _run(["--gpus", 2, "--rendezvous-only"], env={"CLOUDSC2_BENCH_DEADLINE_S": "0.01"}, timeout=120)
    assert r.returncode != 0 and "deadline" in r.stderr, r.stderr[-2000:]


def test_eight_ranks_rendezvous_on_the_cpu():
    """BASELINE configs[4]'s launch shape (8 ranks, one node) as far as a GPU-less box can take it: spawn, rendezvous, the
    reference's split of the columns (dwarf_cloudsc.F90:57-69), the MAX-reduce of the verdict norms
    (cloudsc_driver_tl_mod.F90:125, cloudsc_driver_ad_mod.F90:107 carried across ranks), ONE relayed line."""
    r = _run(["--gpus", 8, "--rendezvous-only", "--ngptot", 160000], timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["stage"] == "final"
    assert d["first_column_per_rank"] == [160000.0 * k for k in range(8)] and d["columns_per_rank"] == 160000
    assert abs(d["verdicts"]["tl_znormg"][0] - 1.8) < 1e-12 and d["verdicts"]["ad_znormg"] == 12.0


def test_a_rank_that_never_answers_fails_the_launch_inside_the_budget():
    """One rank sleeps forever before the rendezvous.  With the budget scaled down to 40 s the others give up the rendezvous
    after budget/3.5 s, exit non-zero, and the parent ends the sleeper: rc != 0 well inside the budget, nothing restarted, the
    reason on stderr.  (Defaults: 420 s and 120 s -- below the 600 s at which the driver kills a run without a record.)"""
    import time

    t0 = time.monotonic()
    r = _run(["--gpus", 3, "--rendezvous-only", "--ngptot", 1000, "--budget-s", 40], env={"CLOUDSC2_BENCH_HANG_RANK": "1"}, timeout=120)
    dt = time.monotonic() - t0
    assert r.returncode != 0, r.stdout
    assert dt < 40.0, dt
    assert "launch of 3 ranks failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip()]


def test_the_default_budget_is_below_the_drivers_limit():
    import bench

    for k in ("CLOUDSC2_BENCH_DEADLINE_S", "CLOUDSC2_DIST_TIMEOUT_S"):
        assert k not in os.environ
    launch, coll = bench.budgets(420.0)
    assert launch <= 450.0 and coll <= 120.0
    launch, coll = bench.budgets(35.0)
    assert launch == 35.0 and coll == 10.0


def test_under_a_launcher_the_timing_line_comes_before_the_full_line():
    """Under `python -m torch.distributed.run` (how the driver starts N > 1) rank 0 prints the timing line as soon as the timing
    is complete and the full line, with the verdicts, last: a run that dies in the verdict legs leaves a parsable record."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--ngptot", "1000"],
                       capture_output=True, text=True, timeout=280, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert [d["stage"] for d in lines] == ["timing", "final"]
    assert "verdicts" not in lines[0] and lines[1]["verdicts"]["ad_znormg"] == 6.0


def test_roofline_fractions_in_every_byte_convention():
    """NL at 160 000 columns in 0.78567 ms (profiles/r03_k_bench_kernel_stats.csv): 0.726 of 8 TB/s on the 28 536 B the launch
    moves, 0.699 on BASELINE.md's 27 440 B; the adjoint's 0.567 algorithmic is 0.68 in counter bytes."""
    import bench

    f = bench.roofline_fractions(0.78567, 160000, 28536, 27440, 4.662e9)
    assert abs(f["frac"] - 0.7264) < 5e-4 and abs(f["frac_kernel_only"] - 0.6985) < 5e-4
    assert abs(f["achieved"] - 5811.3) < 1.0 and f["bytes_per_column_kernel_only"] == 27440
    assert abs(f["frac_actual_bytes"] - 4.662e9 / 0.78567e-3 / 8e12) < 1e-12
    a = bench.roofline_fractions(3.0195502916971844, 160000, 85608, None, 16420340391.266705)
    assert abs(a["frac"] - 0.5670) < 5e-4 and abs(a["frac_actual_bytes"] - 0.6797) < 5e-4 and "frac_kernel_only" not in a
    assert bench.roofline_fractions(1.0, 1000, 8, None, None)["frac_actual_bytes"] is None


def test_the_adjoints_design_floor():
    """CLOUDSC2AD moves 85 608 B per column by SURVEY 8d's count -- if the trajectory survived on chip between the forward pass
    (cloudsc2ad.F90:366-866) and the reverse pass (:877-1740).  It cannot (17.5 KB per column), so the reverse pass reads the
    2 193 trajectory-input doubles a second time: 103 152 B is what a two-pass adjoint must move, and the line carries the fraction
    against that figure next to the algorithmic one.  BENCH_r04's AD: 2.8126 ms at 160 000 columns."""
    import bench
    import dwarf_p_cloudsc2_tl_ad_amd as c2

    assert c2.bytes_per_column(137, "ad", 8) == 85608
    assert c2.bytes_per_column(137, "ad_design_floor", 8) == 85608 + 8 * (138 + 15 * 137) == 103152
    assert c2.bytes_per_column(137, "ad_design_floor", 4) == 103152 // 2
    f = bench.ad_design_floor(c2, 137, 2.8126, 160000)
    assert f["bytes_per_column_design_floor"] == c2.bytes_per_column(137, "ad_design_floor")
    if c2.binding.REAL_BYTES == 8:
        assert abs(f["frac_design_floor"] - 103152 * 160000 / 2.8126e-3 / 8e12) < 1e-12 and abs(f["frac_design_floor"] - 0.7335) < 5e-4
        assert abs(bench.roofline_fractions(2.8126, 160000, 85608)["frac"] - 0.6087) < 5e-4  # the same launch against SURVEY's figure


def test_more_ranks_than_gpus_is_refused_before_anything_starts():
    import torch

    if torch.cuda.device_count() >= 64:
        import pytest

        pytest.skip("a box with 64 GPUs")
    r = _run(["--gpus", 64, "--steps", 1], env={"CLOUDSC2_DIST_BACKEND": ""})
    assert r.returncode == 2 and "nothing started" in r.stderr, r.stderr[-2000:]


def test_pmc_traffic_is_chosen_by_content_not_by_file_name():
    """The fp64 headline line once carried the fp32 library's traffic because of a file-name filter."""
    import bench

    pdir = os.path.join(ROOT, "profiles")
    for rb, per_col in ((8, 28536), (4, 14268)):
        for n in (160000, 1048576):
            algo = per_col * n
            traffic, info = bench.select_pmc_traffic(pdir, "nl", n, rb, algo)
            assert traffic is not None and "same launch size" in info["traffic_source"], info
            assert algo <= traffic < 1.1 * algo, (rb, n, traffic / algo)
            d = json.load(open(os.path.join(ROOT, info["traffic_source"].split(" ")[0])))
            assert d["real_bytes"] == rb and d["ngptot"] == n
    # a pass taken with other algorithmic bytes (an older form of the kernel) is stale: null, never a wrong figure
    traffic, info = bench.select_pmc_traffic(pdir, "nl", 160000, 8, 20000 * 160000)
    assert traffic is None and info["traffic_source"] is None
    assert bench.select_pmc_traffic(os.path.join(ROOT, "no_such_dir"), "nl", 160000, 8, 1)[0] is None


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
    r = _run(["--self-tests", "--ngptot", 100])  # the two self-tests on a resident state: the same refusal
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
    r = _run(["--host-driver-rate", "--ngptot", 100])  # and the host-array driver's rate
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)


def test_the_pmc_parser_reproduces_the_committed_traffic_from_the_raw_counters(tmp_path):
    """bench.py measures its HBM traffic through tools/pmc_parse.traffic (two rocprofv3 --pmc passes, dispatches attributed by order
    against tools/pmc_plan.py, read / write factors calibrated on the SATUR dispatch).  The parser on the round's committed RAW counter
    files gives the committed figures exactly: NL 1.021, TL 1.007, AD 1.198 x the algorithmic bytes, read factor 1.9996 (gfx950
    halves FETCH_SIZE for wide coalesced reads)."""
    import shutil

    from tools import pmc_parse

    raw = os.path.join(ROOT, "profiles", "r05_f_pmc_raw")
    for n in (160000, 1048576):
        for w in ("fetch", "write"):
            d = tmp_path / f"{w}_{n}" / "x"
            d.mkdir(parents=True)
            shutil.copy(os.path.join(raw, f"{w}_{n}_counter_collection.csv"), d / "1_counter_collection.csv")
        got = pmc_parse.traffic(str(tmp_path / f"fetch_{n}"), str(tmp_path / f"write_{n}"), n, 8)
        want = json.load(open(os.path.join(ROOT, "profiles", f"r05_f_{n}_pmc_traffic.json")))
        assert got == want, n
        k = got["kernels"]
        assert abs(got["calibration"]["read_factor"] - 2.0) < 2e-3 and abs(got["calibration"]["write_factor"] - 1.0) < 2e-3
        assert 1.01 < k["nl"]["traffic_over_algorithmic"] < 1.03 and 1.0 < k["tl"]["traffic_over_algorithmic"] < 1.02
        assert 1.18 < k["ad"]["traffic_over_algorithmic"] < 1.21
        # the adjoint's excess IS the second read of the trajectory inputs: traffic ~ the design floor of 103 152 B per column
        assert abs(k["ad"]["traffic_bytes"] / n / 103152 - 1.0) < 0.01, k["ad"]["traffic_bytes"] / n
