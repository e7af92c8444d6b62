"""bench.py itself on the GPU box: the one-GPU line's contract fields, and the N > 1 code path rehearsed with two ranks that share
the one GPU (gloo for the collectives: RCCL refuses two ranks on one device) -- everything the first real 8-GPU run executes
except RCCL itself: spawn, rendezvous, the timed region with its barriers, the MAX over ranks, the per-rank gathers, the timing
line before the verdict legs, the Taylor / adjoint legs on each rank's sub-range, the reductions, the full line last."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *map(str, args)], capture_output=True, text=True,
                          timeout=timeout, env=e, cwd=ROOT)


def test_one_gpu_line_carries_both_byte_conventions():
    r = _bench(["--steps", 20, "--warmup", 3, "--no-companions", "--no-cpu-baseline", "--ngptot", 65536])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["unit"] == "columns/s" and d["dtype"] == "f64" and "stage" not in d
    rf = d["roofline"]
    assert rf["bytes_per_column"] == 28536 and rf["bytes_per_column_kernel_only"] == 27440
    assert abs(rf["frac_kernel_only"] / rf["frac"] - 27440 / 28536) < 1e-12
    assert abs(rf["frac"] - 28536 * 65536 / (rf["kernel_ms_avg"] * 1e-3) / 8e12) < 1e-9
    assert 0.2 < rf["frac"] < 1.0 and d["ms_per_step"] >= rf["kernel_ms_avg"] * 0.999


def test_two_ranks_on_one_gpu_rehearse_the_multi_gpu_path(tmp_path):
    r = _bench(["--gpus", 2, "--steps", 10, "--warmup", 2, "--ngptot", 16384, "--budget-s", 300],
               env={"CLOUDSC2_DIST_BACKEND": "gloo", "CLOUDSC2_BENCH_LOGDIR": str(tmp_path)})
    assert r.returncode == 0, r.stderr[-3000:] + open(tmp_path / "bench_rank0.err").read()[-3000:]
    relayed = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(relayed) == 1  # the parent relays ONE line: the last rank 0 printed
    d = json.loads(relayed[0])
    assert d["stage"] == "final" and d["n_gpus"] == 2 and d["scaling"] == "weak"
    # rank 0's own stdout: the timing line first (complete contract, no verdicts), the full line last
    own = [json.loads(ln) for ln in open(tmp_path / "bench_rank0.out").read().splitlines() if ln.startswith("{")]
    assert len(own) == 2 and own[0]["stage"].startswith("timing") and "verdicts" not in own[0] and own[1] == d
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "config", "roofline"):
        assert own[0][k] == d[k], k
    assert abs(d["value"] - 2 * 16384 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # whole job: both ranks' columns / max time
    per = d["roofline"]["placement_per_rank"]
    assert [p["rank"] for p in per] == [0, 1] and all(p["kernel_ms_avg"] > 0 for p in per)
    assert per[0]["candidates"] == 1  # two ranks share the device: nothing is searched (cloudsc2_alloc.inc)
    assert len(d["roofline"]["kernel_ms_avg_per_rank"]) == 2
    v = d["verdicts"]
    assert v["backend"] == "gloo" and v["tl_passed"] and v["ad_ok"] and len(v["tl_znormg"]) == 10, v
    # BASELINE configs[4] names NL + TL + AD: the N > 1 line carries the other two kernels, timed in-process on every rank with the
    # headline's protocol, and rank 0's CPU baseline
    ck = d["companion_kernels"]
    assert "error" not in ck, ck
    for kind, bpc in (("tl", 57072), ("ad", 85608)):
        c = ck[kind]
        assert c["bytes_per_column"] == bpc and c["unit"] == "columns/s" and c["nproma"] == 128 and c["steps"] >= 5
        assert len(c["kernel_ms_avg_per_rank"]) == 2 and abs(c["kernel_ms_avg"] - max(c["kernel_ms_avg_per_rank"])) < 1e-5 and c["kernel_ms_avg"] > 0
        assert abs(c["value"] - 2 * 16384 / (c["ms_per_step"] * 1e-3)) < 1e-6 * c["value"]
        assert abs(c["frac"] - bpc * 16384 / (c["kernel_ms_avg"] * 1e-3) / 8e12) < 1e-9
    assert ck["ad"]["bytes_per_column_design_floor"] == 103152
    assert abs(ck["ad"]["frac_design_floor"] / ck["ad"]["frac"] - 103152 / 85608) < 1e-9
    assert "companion_kernels" not in own[0]  # the timing line went out before them
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 1e4 and cb["cores"] >= 1, cb


def test_the_default_line_carries_the_nproma_sweep_and_the_adjoints_floor():
    """BASELINE.json configs[1] names an NPROMA sweep 32-256: the driver-run line carries NL at all four blockings (children with their
    own placed state), and the AD companion carries the design floor next to the algorithmic fraction."""
    r = _bench(["--steps", 50, "--warmup", 3, "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    sw = d["nproma_sweep"]
    assert list(sw) == ["32", "64", "128", "256"], sw
    for npr, e in sw.items():
        assert "error" not in e and e["kernel_ms_avg"] > 0 and abs(e["frac"] - 28536 * 160000 / (e["kernel_ms_avg"] * 1e-3) / 8e12) < 1e-9, (npr, e)
    assert sw["128"]["kernel_ms_avg"] == d["roofline"]["kernel_ms_avg"] and sw["128"]["source"].startswith("this run")
    ad = d["companion_kernels"]["ad"]
    assert ad["bytes_per_column"] == 85608 and ad["bytes_per_column_design_floor"] == 103152
    assert abs(ad["frac_design_floor"] - 103152 * 160000 / (ad["kernel_ms_avg"] * 1e-3) / 8e12) < 1e-9 and ad["frac_design_floor"] > ad["frac"]
    assert "bytes_per_column_design_floor" not in d["companion_kernels"]["tl"]
    # the HBM traffic of the line is measured in this very run (two rocprofv3 --pmc child passes), not quoted from an earlier one
    rf = d["roofline"]
    assert rf["traffic_source"].startswith("measured in this run"), rf.get("traffic_measurement_failed", rf["traffic_source"])
    assert 1.0 <= rf["traffic_over_algorithmic"] < 1.06 and abs(rf["traffic"] - rf["traffic_read_bytes"] - rf["traffic_write_bytes"]) < 1.0
    assert abs(rf["frac_actual_bytes"] - rf["traffic"] / (rf["kernel_ms_avg"] * 1e-3) / 8e12) < 1e-9
    assert "profiles/" in rf["traffic_committed_pass"]["traffic_source"]
    assert 1.0 <= d["companion_kernels"]["tl"]["traffic_over_algorithmic"] < 1.05 and 1.15 < ad["traffic_over_algorithmic"] < 1.25
    assert ad["traffic_source"].startswith("measured in this run")
    # guards against a gross performance regression (boxes of the pool measure up to 12 % apart: measured 0.733-0.747 NL, 0.754-0.773 at
    # 1 M columns, 0.72-0.73 TL, 0.73-0.76 of the adjoint's floor; north_star's bar is 0.70 for NL at >= 1 M columns)
    assert rf["frac"] >= 0.65 and d["target_config"]["frac"] >= 0.66, (rf["frac"], d["target_config"]["frac"])
    assert d["companion_kernels"]["tl"]["frac"] >= 0.63 and ad["frac_design_floor"] >= 0.63, (d["companion_kernels"]["tl"]["frac"], ad["frac_design_floor"])
    # BASELINE configs[2] and [3] at their own sizes, in the same line
    bc = d["baseline_configs_2_3"]
    assert "error" not in bc, bc
    t = bc["configs[2] Taylor test, NGPTOT=100, NPROMA=1"]
    a = bc["configs[3] adjoint test, NGPTOT=16384, NPROMA=128"]
    assert t["passed"] and len(t["ratios"]) == 10 and abs(t["ratios"][5] - 1.0) < 1e-4 and t["kernel_ms"] > 0
    assert a["passed"] and a["to_1e-12"] and a["identity_relative"] < 1e-12


def test_host_array_driver_rate_is_reported_beside_the_value():
    """What an unchanged caller of CLOUDSC_DRIVER gets (host arrays through cloudsc2_nl_run, PCIe included) -- a companion of the
    line, never `value`."""
    r = _bench(["--host-driver-rate", "--ngptot", 32768])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["ngptot"] == 32768 and d["columns_per_s"] > 1e5 and d["ms_per_call"] > d["kernel_ms_sum_over_slabs"] > 0.0
    assert "cloudsc2_nl_run" in d["entry_point"] and "value" not in d
