#!/usr/bin/env python
"""bench.py -- CLOUDSC2 NL hot path (SATUR + CLOUDSC2 over all NPROMA blocks) on N MI355X GPUs.

One "step" = one pass of the NL kernel over this rank's NGPTOT=160000 columns x 137 levels (fp64), inputs resident
in HBM.  Weak scaling: every rank owns its own 160000-column sub-range of the global columns (the reference's MPI
split); there is no collective in the data path.  Prints ONE JSON line (rank 0).  `--gpus N` without a launcher starts
the N ranks itself; under `python -m torch.distributed.run` it uses the launcher's ranks.

N > 1 (first exercised on hardware by the driver's scaling run): the time of a rank is its own K steps (synchronize on both sides,
a barrier before and after), `value` = all ranks' columns / the MAX over ranks.  Rank 0 prints the timing line -- every field of the
contract -- as soon as that MAX is known, and the full line (the same plus the self-tests' verdicts reduced over the ranks) LAST, so
that a run that ends in the verdict legs still leaves its measurement on stdout; `python bench.py --gpus N` (own launcher) relays
only the last line.  `--budget-s` (420) bounds the whole launch and, by a fraction of at most 120 s, every rendezvous / collective.

The state is the first and only allocation of the process and comes from the library's allocator, which places it
(cloudsc2_device_malloc_state: candidate allocations judged by the NL sweep itself; profiles/r02_hbm_placement.md): bench.py does
not search.  `roofline.unplaced_first_allocation`
is the same measurement in a fresh process with the placement switched off.

The line also carries `cpu_baseline` (the reference on rank 0's host cores) and, at N=1, from child processes run
after the measurement: `companion_kernels` (the same bench for TL and AD -- BASELINE.json's metric names all three; AD with its
design floor, ad_design_floor), `nproma_sweep` (NL at NPROMA 32 / 64 / 128 / 256, BASELINE configs[1]), `roofline.traffic` measured in this run (two `rocprofv3 --pmc` child passes), `target_config` (NL at
1 048 576 columns, north_star's target), `host_array_driver` (the PCIe-inclusive rate of the reference-signature path, never `value`)
`self_tests` (the Taylor test and the adjoint test on a resident state of the same size: verdicts and kernel time) and
`baseline_configs_2_3` (BASELINE configs[2] and [3] at their own sizes: the Taylor test at 100 columns, the adjoint test at 16 384).  At N > 1
(BASELINE configs[4]: NL + TL + AD on the node) `companion_kernels` comes from every rank timing TL and AD IN-PROCESS on its own
columns after the timing line is out -- same protocol, value = all ranks' columns over the MAX-over-ranks time, per-rank kernel times
-- and `cpu_baseline` from rank 0 after the last collective, while --budget-s has room.  None of this is inside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
SETTLE_LAUNCHES = 15   # untimed launches after the --warmup ones (the GPU's clocks settle; reported in the line)


def select_pmc_traffic(pdir, kernel, ngptot, real_bytes, algorithmic_bytes):
    """HBM traffic of one launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json, written by
    tools/pmc_parse.py).  The pass is chosen by its CONTENT -- `real_bytes` (precision), `ngptot` (launch size), the kernel's
    entry and the algorithmic bytes it was taken with (a pass of an older form of the kernel is stale) -- never by its file
    name; the newest round's file wins among equals.  A figure far below the algorithmic bytes is impossible for this path and
    is rejected (the fp32 library's pass on an fp64 line would read 0.5; the reverse sweep of the adjoint alone legitimately
    reads 0.96: the compiler drops the loads of two adjoint planes that only feed the compiled-out evaporation branch, which
    SURVEY's per-plane count includes).  Returns (bytes or None, dict of fields for the line)."""
    cands = []
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith("pmc_traffic.json"):
            continue
        try:
            d = json.load(open(os.path.join(pdir, f)))
            k = d["kernels"][kernel]
            n = int(d["ngptot"])
            if int(d["real_bytes"]) != int(real_bytes):
                continue
            if abs(float(k["algorithmic_bytes"]) / n * ngptot - algorithmic_bytes) > 1e-6 * algorithmic_bytes:
                continue  # taken when the kernel moved other bytes
            cands.append((n == int(ngptot), f, float(k["traffic_bytes"]) / n * ngptot))
        except (KeyError, ValueError, TypeError, OSError):
            continue  # older files without the fields, other kernels
    if not cands:
        return None, {"traffic_source": None}
    cands.sort(key=lambda c: (c[0], c[1]))  # same launch size first, then the newest file name (rNN_ prefixes sort by round)
    same, fname, traffic = cands[-1]
    info = {"traffic_source": f"profiles/{fname} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run, "
                              f"{'same' if same else 'another'} launch size, scaled per column; not measured in this run)"}
    if traffic < 0.9 * algorithmic_bytes:
        info["traffic_rejected"] = f"{traffic:.4g} B is below the algorithmic {algorithmic_bytes:.4g} B"
        return None, info
    return traffic, info


def measure_pmc_traffic(ngptot, precision, real_bytes, timeout_s=150.0):
    """HBM traffic per launch of NL / TL / AD measured NOW, as MI355X_MICROARCH.md's HBM section prescribes: two child runs of
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace flags, the program itself after `--`) over
    tools/pmc_workload.py at this launch size, attributed by dispatch order and calibrated on the SATUR dispatch (tools/pmc_parse.py).
    Returns the parsed dict; raises when the profiler is missing, a pass fails or runs past `timeout_s` (the caller then falls back
    to the committed pass).  Each child is its own process group, killed as a whole at the deadline."""
    import shutil
    import signal
    import subprocess
    import tempfile

    from tools import pmc_parse

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="cloudsc2_pmc_", dir="/tmp")
    env = {**os.environ, "CLOUDSC2_PLACE": "0", "PMC_NGPTOT": str(ngptot), "CLOUDSC2_PRECISION": precision, "TMPDIR": "/tmp"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cmd = [rp, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, sub), "--", "python3", os.path.join(ROOT, "tools", "pmc_workload.py")]
            p = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                outp, _ = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                raise RuntimeError(f"rocprofv3 --pmc {counter}: no end after {timeout_s:g} s") from None
            if p.returncode != 0 or "pmc workload done" not in outp:
                raise RuntimeError(f"rocprofv3 --pmc {counter} failed (rc {p.returncode}): {outp[-300:]}")
        return pmc_parse.traffic(os.path.join(tmp, "fetch"), os.path.join(tmp, "write"), ngptot, real_bytes)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline_fractions(kernel_ms, ngptot, bytes_per_column, kernel_only_bytes_per_column=None, traffic_bytes=None, peak_gbs=HBM_PEAK_GBS):
    """The fractions of the HBM peak one launch of `kernel_ms` reaches, in every byte convention the line carries:
      frac               algorithmic bytes of what the launch moves (NL: 28 536 B/column at NLEV 137, i.e. with the driver's
                         CLD(:,:,NCLV)=0 plane the launch really writes -- the PMC write bytes agree);
      frac_kernel_only   NL only: BASELINE.md section 2 / SURVEY 8d's first figure, 27 440 B/column, the CLOUDSC2 dummies alone;
      frac_actual_bytes  the HBM bytes the counters saw (rocprofv3 FETCH_SIZE + WRITE_SIZE of the committed PMC pass) over the same
                         time -- for the adjoint 1.2 x the algorithmic bytes (the reverse sweep re-reads the trajectory planes), so
                         0.57 algorithmic is 0.68 of the peak in bytes actually moved.
    Pure arithmetic (tests/test_bench_launch.py)."""
    t = kernel_ms * 1e-3
    ach = bytes_per_column * ngptot / t / 1e9
    out = {"achieved": ach, "frac": ach / peak_gbs}
    if kernel_only_bytes_per_column is not None:
        out["bytes_per_column_kernel_only"] = int(kernel_only_bytes_per_column)
        out["frac_kernel_only"] = kernel_only_bytes_per_column * ngptot / t / 1e9 / peak_gbs
    out["frac_actual_bytes"] = (traffic_bytes / t / 1e9 / peak_gbs) if traffic_bytes else None
    return out


def budgets(budget_s):
    """The two deadlines of an N-rank run from ONE figure: the whole launch must end (or be killed with a record) inside `budget_s`,
    a rendezvous or collective that gets no answer fails after a fraction of it.  Environment overrides stay for tests."""
    launch = float(os.environ.get("CLOUDSC2_BENCH_DEADLINE_S", budget_s))
    coll = float(os.environ.get("CLOUDSC2_DIST_TIMEOUT_S", max(5.0, min(120.0, budget_s / 3.5))))
    return launch, coll


def effective_cores() -> int:
    """CPU share of this process: cgroup quota if one is set, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return int(os.environ.get("CLOUDSC2_CPU_THREADS", n))


def cpu_baseline(tab, prm, nproma, ngptot, budget_s=20.0):
    """The checker timed on the host cores (reported baseline, not the target).  kind "reference" = the unmodified
    reference driver+kernels (oracle/_ref), else the C port.  Same workload, bounded to ~budget_s of CPU work."""
    import dwarf_p_cloudsc2_tl_ad_amd as c2
    from oracle import refcall

    cores = effective_cores()
    st = c2.state_from_table(tab, nproma, ngptot)
    single = c2.binding.SINGLE
    if refcall.have_ref(single=single):
        lib, kind = refcall.RefLib(single=single), "reference"
        lib.set_params(prm.doubles30(), prm.ceta_array())
        os.environ.setdefault("OMP_SCHEDULE", "static")
        arrays = st.driver_arrays()
        run = lambda: lib.driver(0, cores, nproma, st.nlev, ngptot, st.ptsphy, arrays)  # noqa: E731
        run4 = lambda: lib.driver(0, min(4, cores), nproma, st.nlev, ngptot, st.ptsphy, arrays)  # noqa: E731
    elif refcall.have_oracle() and not single:
        lib, kind = refcall.OracleLib(), "port"
        lib.set_params(prm.doubles30(), prm.ceta_array())
        import ctypes as C

        dp = C.POINTER(C.c_double)
        lib.lib.oracle_driver_nl.argtypes = [C.c_int] * 4 + [C.c_double] + [dp] * 18
        ptrs = [a.ctypes.data_as(dp) for a in st.driver_arrays()]
        run = lambda: lib.lib.oracle_driver_nl(cores, nproma, st.nlev, ngptot, st.ptsphy, *ptrs)  # noqa: E731
        run4 = lambda: lib.lib.oracle_driver_nl(min(4, cores), nproma, st.nlev, ngptot, st.ptsphy, *ptrs)  # noqa: E731
    else:
        return None
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference driver prints its timing table on stderr
    try:
        run()  # warm-up (page faults)
        times = []
        t_end = time.perf_counter() + budget_s
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 20):
            t0 = time.perf_counter()
            run()
            times.append(time.perf_counter() - t0)
        times4 = []  # the reference README's own invocation: dwarf-cloudsc2-nl 4 160000 32 (README.md:49)
        for _ in range(4):
            t0 = time.perf_counter()
            run4()
            times4.append(time.perf_counter() - t0)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    best = float(np.median(times))
    return {"value": ngptot / best, "unit": "columns/s", "cores": cores, "kind": kind,
            "numomp4_value": ngptot / float(np.median(times4[1:])),
            "sample": f"NL, {ngptot} columns x 137 levels, NPROMA {nproma}, median of {len(times)} full passes "
                      f"({best * 1e3:.0f} ms each), OMP_SCHEDULE=static"}


def device_info(torch, dev):
    """What the box is (boxes of one pool measure up to 12 % apart with the same binary)."""
    try:
        p = torch.cuda.get_device_properties(dev)
        info = {"name": p.name, "compute_units": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2**30, 1)}
        for k in ("clock_rate", "memory_clock_rate", "gcnArchName"):
            if hasattr(p, k):
                info[k] = getattr(p, k)
        return info
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus() -> int:
    """GPUs this process could use, WITHOUT initialising HIP (device_count does not, on this image)."""
    try:
        import torch

        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


def _tail(path, n=2000):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            size = f.tell()
            f.seek(max(0, size - n))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def spawn_ranks(ngpus: int, argv, rehearsal: bool = False, budget_s: float = 420.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) of this same script and
    relay rank 0's JSON line.  Runs before anything in this process has touched the GPU; the children get
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exactly as `python -m torch.distributed.run --nproc-per-node N` would set
    them, so both ways of starting the bench run the same code.

    The launch cannot die silently (the reference's ranks are MPI processes: one failing rank aborts the job,
    dwarf_cloudsc.F90:57-69): every rank's stderr goes to bench_rank<r>.err, all children are polled, the first non-zero
    exit terminates the others within seconds and its stderr tail is relayed, an overall deadline bounds the run, and nothing
    is restarted or re-exec'ed."""
    import subprocess
    import tempfile

    if not rehearsal and os.environ.get("CLOUDSC2_DIST_BACKEND", "") != "gloo":
        have = visible_gpus()
        if have < ngpus:
            print(f"bench.py: --gpus {ngpus} but only {have} GPU(s) are visible; nothing started "
                  "(CLOUDSC2_DIST_BACKEND=gloo rehearses the multi-rank path on fewer GPUs)", file=sys.stderr)
            return 2
    logdir = os.environ.get("CLOUDSC2_BENCH_LOGDIR") or tempfile.mkdtemp(prefix="cloudsc2_bench_")
    os.makedirs(logdir, exist_ok=True)
    launch_s, coll_s = budgets(budget_s)
    deadline = time.monotonic() + launch_s
    env = dict(os.environ, WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_WORLD_SIZE=str(ngpus))
    env.setdefault("CLOUDSC2_DIST_TIMEOUT_S", str(coll_s))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    procs, errs = [], []
    out0 = os.path.join(logdir, "bench_rank0.out")
    for r in range(ngpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        errs.append(os.path.join(logdir, f"bench_rank{r}.err"))
        with open(errs[r], "wb") as ferr, open(out0 if r == 0 else os.devnull, "wb") as fout:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e, stdout=fout, stderr=ferr))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    failed, why = None, None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed, why = bad, "exit code"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed, why = [(r, None) for r, rc in enumerate(rcs) if rc is None], f"still running at the deadline ({launch_s:g} s: --budget-s / CLOUDSC2_BENCH_DEADLINE_S)"
            break
        time.sleep(0.2)
    if failed:
        stop_all()
    lines = [ln for ln in _tail(out0, 1 << 20).splitlines() if ln.strip().startswith("{")]
    if lines:  # the last line rank 0 got out: the full one, or the timing line of a run whose verdict legs never finished
        print(lines[-1], flush=True)
    if failed or not lines:
        print(f"bench.py: launch of {ngpus} ranks failed -- (rank, exit code) {failed}: {why}; per-rank stderr in {logdir}", file=sys.stderr)
        for r, _ in (failed or [(0, None)])[:2]:
            tail = _tail(errs[r]).strip()
            if tail:
                print(f"---- rank {r} stderr (tail) ----\n{tail}", file=sys.stderr)
        return 1
    return 0


def build_step(c2, args, tab, prm, kernel, dev, stream, col0):
    """One kernel of the path on this rank's columns, ready to be timed: (state, step(), algorithmic bytes per column, what must stay
    alive, the kernel's name, the allocator's placement record).

    The state: tiled on the device from the 100-column table (cloudsc2_expand_launch: no host copy exists) into ONE arena
    from the library's allocator, which places it (cloudsc2_device_malloc_state: candidate allocations spanning 96 GiB, judged by
    the NL sweep itself for a state alone, by two generic streams when perturbation sets follow; profiles/r02_hbm_placement.md).
    What any caller of the C ABI gets, no search here.
    TL / AD: the perturbation set (increments + outputs) and the adjoint's carry plane live in the SAME allocation as the state,
    as in the library's own test drivers (measured: TL 1.65 vs 1.66-1.72 ms, AD 3.04 vs 3.18-3.22 ms for a separate allocation)."""
    nbk = (args.ngptot + args.nproma - 1) // args.nproma
    nlev_t = tab["PT"].shape[0]
    reserve = 0
    if kernel != "nl":
        reserve = c2.FlatFields.pair_bytes(nbk, nlev_t, args.nproma) + (nbk * nlev_t * args.nproma * c2.binding.REAL_BYTES + 4096 if (kernel == "ad" and args.levapls2) else 0)
    ds = c2.DeviceState.from_table(tab, args.nproma, args.ngptot, dev, start=col0, reserve=reserve)
    placement = dict(getattr(ds.arena, "info", {}))
    nlev = ds.nlev
    if kernel == "nl":
        step, bpc, keep = (lambda: ds.nl(prm, stream)), c2.bytes_per_column(nlev, "nl_driver"), ds
        kname = "nl_kernel<F> (SATUR + CLOUDSC2 fused; fast math, no evaporation branch, 32-bit offsets when buffers < 4 GiB)"
    else:
        ds.satur(prm, stream)
        inc, dout = c2.FlatFields.pair(ds.nb, ds.nlev, ds.nproma, dev, arena=ds.arena)  # increments + TL outputs
        ds.increments(zero_supsat=(kernel == "ad"), into=inc)
        if kernel == "tl":
            step, bpc, keep = (lambda: ds.tl(prm, inc, dout, stream)), c2.bytes_per_column(nlev, "tl"), (ds, inc, dout)
            kname = ("tl_kernel<C2F_QSAT|C2F_TRAJ> (CLOUDSC2TL: trajectory evaluated in the sweep, its ten outputs and the ten TL outputs stored; "
                     "launches of a few partial rounds of workgroups are paced, CLOUDSC2_PACE=0 switches that off)")
        else:
            ds.tl(prm, inc, dout, stream)  # leaves the trajectory outputs (PFPLSL5 / PFPLSN5) in the state
            # the cover-checkpoint plane exists only with the evaporation branch (its one reader)
            scratch = ds.arena.take((ds.nb, ds.nlev, ds.nproma)) if args.levapls2 else None
            step = lambda: ds.ad(prm, inc, dout, scratch, stream, assign=args.ad_assign, sweep=args.ad_sweep)  # noqa: E731
            bpc = c2.bytes_per_column(nlev, "ad" if args.ad_sweep == "both" else "ad_reverse")
            if args.levapls2:  # + the checkpoint plane: written by the forward sweep, read by the reverse sweep
                bpc += c2.bytes_per_column(nlev, "ad_ckpt") // (1 if args.ad_sweep == "both" else 2)
            kname = ("ad_kernel<C2F_QSAT> (CLOUDSC2AD: trajectory pass + reverse pass; partial rounds paced)" if args.ad_sweep == "both" else
                     "ad_reverse_kernel<C2F_QSAT> (reverse sweep of CLOUDSC2AD alone; carries from the state's PFPLSL5 / PFPLSN5; partial rounds paced)")
            if args.ad_assign:  # the 16 old input adjoints (15 full-level planes + PAPH's nlev+1) are not read
                bpc -= c2.bytes_per_column(nlev, "ad_old_adjoints")
                kname = kname.replace("<C2F_QSAT>", "<C2F_QSAT|C2F_ASSIGN>") + " [x = A^T y: old input adjoints not read]"
            keep = (ds, inc, dout, scratch)
    return ds, step, bpc, keep, kname, placement


def ad_design_floor(c2, nlev, kernel_ms, ngptot, peak_gbs=HBM_PEAK_GBS, evap=False):
    """CLOUDSC2AD, both sweeps, accumulate form: SURVEY 8d's 85 608 B per column assume the trajectory survives on chip between the
    forward pass (cloudsc2ad.F90:366-866) and the reverse pass (:877-1740).  A 137-level column's trajectory inputs are 17 544 B; a
    CU's 256 resident columns would need 4.5 MB of the 160 KiB LDS.  What a two-pass adjoint must move is therefore 85 608 + the
    second read of the 2 193 trajectory-input doubles = 103 152 B per column (the PMC counters see 1.198 x 85 608 = 102.6 KB), and
    the fraction of the peak against THAT figure is what the kernel can be held to.  Pure arithmetic (tests/test_bench_launch.py)."""
    floor, algo = c2.bytes_per_column(nlev, "ad_design_floor"), c2.bytes_per_column(nlev, "ad")
    if evap:  # + the cover-checkpoint plane the evaporation branch writes in the forward pass and reads in the reverse pass
        floor, algo = floor + c2.bytes_per_column(nlev, "ad_ckpt"), algo + c2.bytes_per_column(nlev, "ad_ckpt")
    return {"bytes_per_column_design_floor": int(floor),
            "frac_design_floor": floor * ngptot / (kernel_ms * 1e-3) / 1e9 / peak_gbs,
            "design_floor": f"{algo} B (SURVEY 8d's count) + the reverse pass's second read of the {(nlev + 1) + 15 * nlev} trajectory-input values: a "
                            f"{nlev}-level trajectory does not survive on chip between the two passes (cloudsc2ad.F90:366-866, :877-1740)"}


def timed_steps(torch, dev, stream, step, steps, warmup, barrier=None):
    """W warm-up + SETTLE_LAUNCHES untimed launches, then exactly `steps` launches between a barrier + synchronize on both sides.
    Returns (wall seconds of the K steps on this rank, per-launch device times from HIP events on the launch stream)."""
    for _ in range(max(warmup, 0)):
        step()
    for _ in range(SETTLE_LAUNCHES):  # after an idle second the GPU needs ~15 launches (12 ms) to reach its steady time again, whatever the memory
        # (tools/settle_series.py: 0.93 0.84 0.86 0.89 0.87 ... 0.82 ms; the same after a 2 s pause); reported as `settle_launches`,
        # next to the W warm-up steps the caller asked for (`warmup_total` = both)
        step()
    torch.cuda.synchronize(dev)
    if barrier:
        barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0  # this rank's K steps, device work included; the closing barrier follows, then the MAX over ranks
    if barrier:
        barrier()
    return elapsed, np.array([a.elapsed_time(b) for a, b in ev])  # per-launch device time on the launch stream


def child_bench(extra_args, env=None, timeout=900):
    """Run this script as a fresh child process (own device memory, own placement) and return its JSON line."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__)] + [str(a) for a in extra_args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=None if env is None else {**os.environ, **env})
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError(f"child bench failed (rc {r.returncode}): {r.stderr[-400:]}")
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ngptot", type=int, default=160000, help="columns per GPU")
    ap.add_argument("--nproma", type=int, default=128)
    ap.add_argument("--kernel", choices=["nl", "tl", "ad"], default="nl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true",
                    help="skip everything appended to the headline line from child processes: TL and AD timings, the 1 M-column "
                         "NL target configuration, the unplaced first allocation")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure the HBM traffic in this run (two rocprofv3 --pmc child passes, ~25 s): `roofline.traffic` then comes "
                         "from the newest committed pass under profiles/")
    ap.add_argument("--precision", choices=["double", "single"], default=os.environ.get("CLOUDSC2_PRECISION", "double"),
                    help="single = the fp32 library (the reference's -DSINGLE build); the headline metric is double")
    ap.add_argument("--ad-assign", action="store_true",
                    help="--kernel ad: the assign form x = A^T y (cloudsc2_ad_launch_assign, what the adjoint test uses) instead of "
                         "CLOUDSC2AD's accumulate form; its algorithmic bytes do not include reading the old input adjoints")
    ap.add_argument("--ad-sweep", choices=["both", "reverse"], default="both",
                    help="--kernel ad: reverse = the reverse sweep alone (cloudsc2_ad_launch_reverse) on the PFPLSL5 / PFPLSN5 an earlier "
                         "sweep left in the state -- the adjoint leg of cloudsc2_ad_symmetry_run")
    ap.add_argument("--levapls2", action="store_true", help="switch the evaporation branch on (off in every shipped config)")
    ap.add_argument("--self-tests", action="store_true",
                    help="no bench line: the reference's two self-tests (CLOUDSC_DRIVER_TL's Taylor test, CLOUDSC_DRIVER_AD's adjoint test) on "
                         "a resident state of --ngptot columns, with their verdicts and the kernel time of the whole driver call")
    ap.add_argument("--host-driver-rate", action="store_true",
                    help="no bench line: the PCIe-inclusive rate of the reference-signature path -- cloudsc2_nl_run, what CLOUDSC_DRIVER binds, "
                         "on host arrays in pageable memory (upload, kernel, download per call); reported beside `value`, never as `value`")
    ap.add_argument("--budget-s", type=float, default=420.0,
                    help="N > 1: everything must be over inside this many seconds (the driver kills a run at 600 s and then nothing is "
                         "written): the launcher's overall deadline, and -- a fraction of it, at most 120 s -- the timeout of the "
                         "rendezvous and of every collective.  CLOUDSC2_BENCH_DEADLINE_S / CLOUDSC2_DIST_TIMEOUT_S override")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="no GPU work: start the ranks, rendezvous, shard the columns, reduce fake verdict norms, print the line "
                         "(CPU rehearsal of the launch path; tests/test_bench_launch.py)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # no launcher: become one (nothing here has touched the GPU yet)
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:], rehearsal=args.rendezvous_only, budget_s=args.budget_s))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    # under any launcher (ours or torch.distributed.run): the collectives' timeout follows the budget unless the environment says otherwise
    os.environ.setdefault("CLOUDSC2_DIST_TIMEOUT_S", str(budgets(args.budget_s)[1]))
    t_start = time.monotonic()
    if os.environ.get("CLOUDSC2_BENCH_HANG_RANK") == os.environ.get("RANK", "0"):  # tests/test_bench_launch.py: a rank that never answers
        time.sleep(3600)
    if os.environ.get("CLOUDSC2_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):  # tests/test_bench_launch.py: a rank that dies at start
        print("bench.py: this rank was told to fail at start (CLOUDSC2_BENCH_FAIL_RANK)", file=sys.stderr)
        raise SystemExit(3)

    import torch

    os.environ["CLOUDSC2_PRECISION"] = args.precision  # read by the package at import: one precision per process
    import dwarf_p_cloudsc2_tl_ad_amd as c2
    from dwarf_p_cloudsc2_tl_ad_amd import dist as c2dist

    single = c2.binding.SINGLE
    fp = "fp32" if single else "fp64"
    variant = ", the -DSINGLE variant" if single else ""
    if args.ad_assign and args.kernel == "ad":
        variant += ", assign form of the adjoint"
    if args.ad_sweep == "reverse" and args.kernel == "ad":
        variant += ", reverse sweep alone"

    if args.self_tests:
        if not c2.device_available():
            raise SystemExit("bench.py needs a HIP device: the CLOUDSC2 engine has no CPU path")
        tab = c2.synthetic_table()
        ceta = c2.ceta_from_table(tab)
        rs = c2.ResidentState.from_table(tab, args.nproma, args.ngptot)
        res = {"ngptot": args.ngptot, "nproma": args.nproma, "dtype": "f32" if single else "f64"}
        t = [rs.tl_taylor(c2.default_params(ceta, lregcl=False)) for _ in range(4)]  # (precise arithmetic: the Taylor driver's default)
        res["taylor_test"] = {"kernel_ms": min(x[3] for x in t[1:]), "passed": bool(t[-1][1]), "penalty": int(t[-1][2]),
                              "ratios": [float(z) for z in t[-1][0]],
                              "what": "SATUR, TL sweep storing the base trajectory, the ten lambdas in one sweep on the lanes of a wave, block sums"}
        a = [rs.ad_symmetry(c2.default_params(ceta, lregcl=True)) for _ in range(4)]
        res["adjoint_test"] = {"kernel_ms": min(x[2] for x in a[1:]), "passed": bool(a[-1][1]), "znormg_in_eps": float(a[-1][0]),
                               "what": "SATUR, TL sweep forming <y,y>, reverse sweep alone (assign form) forming <x0,x_adj> and norm3"}
        print(json.dumps(res), flush=True)
        return

    if args.host_driver_rate:
        if not c2.device_available():
            raise SystemExit("bench.py needs a HIP device: the CLOUDSC2 engine has no CPU path")
        tab = c2.synthetic_table()
        prm = c2.default_params(c2.ceta_from_table(tab))
        st = c2.state_from_table(tab, args.nproma, args.ngptot)
        walls, kms = [], []
        for _ in range(4):  # the first call allocates the workspace (one plain hipMalloc) and pages the host arrays in
            t0 = time.perf_counter()
            k = c2.run_state(prm, st, "nl")
            walls.append(time.perf_counter() - t0)
            kms.append(float(k))
        w = float(np.median(walls[1:]))
        print(json.dumps({"entry_point": "cloudsc2_nl_run (what CLOUDSC_DRIVER binds): host arrays in pageable memory, 17.5 KB up + 12.1 KB down per column, "
                                         "slabs of 16 384 columns with upload / kernel / download overlapped",
                          "ngptot": args.ngptot, "nproma": args.nproma, "ms_per_call": w * 1e3, "columns_per_s": args.ngptot / w,
                          "kernel_ms_sum_over_slabs": float(np.median(kms[1:])), "first_call_ms": walls[0] * 1e3,
                          "note": "PCIe-bound; never `value` -- the resident state is what the kernel's rate needs"}), flush=True)
        return

    if args.rendezvous_only:
        rank, local, world = c2dist.init_process_group("gloo")
        col0, ncols = c2dist.shard(args.ngptot * world, rank, world)
        tl = c2dist.allreduce_max([1.0 + 10.0 ** (-(k + 1)) * (rank + 1) for k in range(10)])
        ad = c2dist.allreduce_max([5.0 + rank])
        per_rank = c2dist.allgather_scalar(float(col0))
        if rank == 0:
            line = {"metric": "rendezvous only (no GPU work)", "n_gpus": world, "rendezvous_only": True,
                    "first_column_per_rank": per_rank, "columns_per_rank": ncols}
            if world > 1:  # the same two-stage protocol as the measurement: the timing line first, the full line last
                print(json.dumps({**line, "stage": "timing"}), flush=True)
                line["stage"] = "final"
            line["verdicts"] = {"tl_znormg": [float(x) for x in tl], "ad_znormg": float(ad[0])}
            print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    rank, local, world = c2dist.init_process_group()
    if not torch.cuda.is_available() or not c2.device_available():
        raise SystemExit("bench.py needs a HIP device: the CLOUDSC2 engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=(args.kernel == "ad"), levapls2=args.levapls2)
    col0 = rank * args.ngptot  # weak scaling: rank r owns global columns [r*NGPTOT, (r+1)*NGPTOT)
    stream = torch.cuda.current_stream(dev)

    ds, step, bpc, keep, kname, placement = build_step(c2, args, tab, prm, args.kernel, dev, stream, col0)
    nlev = ds.nlev

    def barrier():
        if torch.distributed.get_backend() == "nccl":
            torch.distributed.barrier(device_ids=[local])  # the rank's own GPU, not a guess
        else:
            torch.distributed.barrier()

    elapsed, kms = timed_steps(torch, dev, stream, step, args.steps, args.warmup, barrier if world > 1 else None)
    if world > 1:
        elapsed = float(c2dist.allreduce_max([elapsed], dev)[0])  # MAX over ranks (RCCL all-reduce of one double)
    ms_per_step = elapsed / args.steps * 1e3
    total_cols = args.ngptot * world
    value = total_cols / (elapsed / args.steps)

    k_avg = float(kms.mean())
    k_per_rank = c2dist.allgather_scalar(k_avg, dev) if world > 1 else [k_avg]
    # HBM traffic: first the figure of the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, calibrated as the
    # MI355X guide prescribes; tools/pmc_workload.py + tools/pmc_parse.py), scaled per column to this launch (`traffic_source` names
    # the file); the default one-GPU run then MEASURES it itself in two child passes (measure_pmc_traffic, companions below) and
    # replaces the figure -- `traffic_source` says which of the two the line carries.
    tkey = args.kernel
    if args.kernel == "ad":
        tkey = ("ad" if args.ad_sweep == "both" else "ad_reverse") + ("_assign" if args.ad_assign else "")
    traffic, tinfo = select_pmc_traffic(os.path.join(ROOT, "profiles"), tkey, args.ngptot, c2.binding.REAL_BYTES, bpc * args.ngptot)
    # both byte conventions (BASELINE.md section 2 divides NL by the 27 440 B of the CLOUDSC2 dummies alone; the launch also writes
    # the driver's CLD(:,:,NCLV)=0 plane: 28 536 B) and the counters' bytes over the same time
    fr = roofline_fractions(k_avg, args.ngptot, bpc, c2.bytes_per_column(nlev, "nl") if args.kernel == "nl" else None, traffic)
    achieved = fr["achieved"]
    roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": fr["frac"], "traffic": traffic,
                "traffic_over_algorithmic": (traffic / (bpc * args.ngptot)) if traffic else None, **tinfo, "bytes_per_column": bpc,
                **{k: v for k, v in fr.items() if k not in ("achieved", "frac")},
                "algorithmic_bytes": bpc * args.ngptot, "kernel_ms_avg": k_avg, "kernel_ms_min": float(kms.min()),
                "kernel_ms_first_tenth": float(kms[:max(1, len(kms) // 10)].mean()), "kernel_ms_last_tenth": float(kms[-max(1, len(kms) // 10):].mean()),
                "kernel_ms_avg_per_rank": [round(x, 5) for x in k_per_rank],
                "allocation": "first and only state of the process, from cloudsc2_device_malloc_state (placed by the library)"}
    if args.kernel == "ad" and args.ad_sweep == "both" and not args.ad_assign:
        roofline.update(ad_design_floor(c2, nlev, k_avg, args.ngptot, evap=args.levapls2))
    if world > 1:  # every rank's placement next to its kernel time (a slow rank is a slow place or a slow GPU: this tells which)
        pr = {k: c2dist.allgather_scalar(float(placement.get(k, 0.0)), dev) for k in ("candidates", "probe_ms_best", "probe_ms_median", "probe_ms_worst")}
        roofline["placement_per_rank"] = [{"rank": r, "candidates": int(pr["candidates"][r]), "probe_ms_best": pr["probe_ms_best"][r],
                                           "probe_ms_median": pr["probe_ms_median"][r], "probe_ms_worst": pr["probe_ms_worst"][r],
                                           "kernel_ms_avg": round(k_per_rank[r], 5)} for r in range(world)]
    if args.kernel == "nl" and placement.get("candidates", 0) > 1 and not os.environ.get("CLOUDSC2_PLACE_PROBE"):
        # a state alone is judged by the NL sweep itself (zero-filled state in every candidate), so the allocator's probe times ARE
        # kernel times: what the median and the worst candidate of this box would have given
        roofline["candidates"] = {"count": placement["candidates"], "judge": "NL sweep on a zero-filled state in each candidate",
                                  "kernel_ms_chosen": placement["probe_ms_best"], "kernel_ms_median": placement["probe_ms_median"],
                                  "kernel_ms_worst": placement["probe_ms_worst"],
                                  "frac_median_candidate": bpc * args.ngptot / (placement["probe_ms_median"] * 1e-3) / 1e9 / HBM_PEAK_GBS}

    out = {
        "metric": f"CLOUDSC2 {args.kernel.upper()} columns/sec ({fp}, NLEV=137)", "value": value, "unit": "columns/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_launches": SETTLE_LAUNCHES,
        "warmup_total": max(args.warmup, 0) + SETTLE_LAUNCHES, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if single else "f64", "data": "synthetic",
        "config": {"workload": f"CLOUDSC2 {args.kernel.upper()} {fp}, NGPTOT={args.ngptot} columns per GPU, NLEV=137, "
                               f"NPROMA={args.nproma} (BASELINE.json configs[1]{variant})",
                   "ngptot_per_gpu": args.ngptot, "nlev": nlev, "nproma": args.nproma,
                   "parallelism": f"columns sharded over {world} GPU(s), no data-path collective",
                   "device": device_info(torch, dev),
                   "placement": placement},
        "roofline": roofline,
    }
    if world > 1 and rank == 0:
        # The timing is complete here: it goes out NOW, as a line of its own with every field of the contract, before the verdict
        # legs below touch RCCL again -- whatever happens there (a collective that never answers is ended by the watchdog with the
        # process), the measurement is on record.  The full line, with `verdicts`, follows as the LAST line.
        print(json.dumps({**out, "stage": "timing (the full line with the self-tests' verdicts follows; this one stands if it does not)",
                          "seconds_since_start": round(time.monotonic() - t_start, 1)}), flush=True)
    companions = rank == 0 and world == 1 and args.kernel == "nl" and not args.no_companions
    if companions:
        # Everything below comes from fresh child processes started after this process has given its device memory back; none of
        # it is inside the timed region or part of `value`; `--no-companions` skips it.
        del step, keep, ds
        torch.cuda.empty_cache()
        common = ["--no-cpu-baseline", "--no-companions", "--nproma", args.nproma, "--precision", args.precision] + \
            (["--levapls2"] if args.levapls2 else [])
        # (1) BASELINE.json's metric names NL/TL/AD: the same bench for the other two kernels
        comp = {}
        for kind in ("tl", "ad"):
            try:
                d = child_bench(["--kernel", kind, "--steps", 30, "--warmup", 5, "--ngptot", args.ngptot] + common)
                comp[kind] = {"value": d["value"], "unit": d["unit"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                              "bytes_per_column": d["roofline"]["bytes_per_column"], "frac": d["roofline"]["frac"],
                              "traffic": d["roofline"]["traffic"], "traffic_over_algorithmic": d["roofline"]["traffic_over_algorithmic"],
                              "frac_actual_bytes": d["roofline"]["frac_actual_bytes"], "traffic_source": d["roofline"].get("traffic_source"),
                              "kernel": d["roofline"]["kernel"],
                              "placement": d["config"]["placement"]}
                for k in ("bytes_per_column_design_floor", "frac_design_floor", "design_floor"):
                    if k in d["roofline"]:
                        comp[kind][k] = d["roofline"][k]
            except Exception as e:  # noqa: BLE001  (never let the companions break the headline line)
                comp[kind] = {"error": repr(e)}
        out["companion_kernels"] = comp
        # (1a) HBM traffic measured in THIS run (the line used to carry a committed pass only): replaces the figures
        # chosen from profiles/ above for NL and for the two companions; on any failure the committed figures stay, and say so
        if not args.no_pmc and not args.levapls2:
            try:
                t0 = time.perf_counter()
                pm = measure_pmc_traffic(args.ngptot, args.precision, c2.binding.REAL_BYTES)
                how = (f"measured in this run: two child passes of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_workload.py at {args.ngptot} "
                       f"columns (first plain hipMalloc of their processes), calibrated on the SATUR dispatch (read factor "
                       f"{pm['calibration']['read_factor']:.4f}); {time.perf_counter() - t0:.0f} s")

                def apply(dst, key, kernel_ms):
                    k = pm["kernels"][key]
                    dst["traffic_committed_pass"] = {"traffic": dst.get("traffic"), "traffic_source": dst.get("traffic_source")}
                    dst["traffic"] = k["traffic_bytes"]
                    dst["traffic_over_algorithmic"] = k["traffic_over_algorithmic"]
                    dst["frac_actual_bytes"] = k["traffic_bytes"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    dst["traffic_source"] = how
                    dst["traffic_read_bytes"], dst["traffic_write_bytes"] = k["read_bytes"], k["write_bytes"]

                if abs(pm["kernels"]["nl"]["algorithmic_bytes"] - bpc * args.ngptot) < 1e-6 * bpc * args.ngptot:
                    apply(out["roofline"], "nl", k_avg)
                for kind in ("tl", "ad"):
                    if "kernel_ms_avg" in comp.get(kind, {}):
                        apply(comp[kind], kind, comp[kind]["kernel_ms_avg"])
            except Exception as e:  # noqa: BLE001
                out["roofline"]["traffic_measurement_failed"] = repr(e)[:400]
        # (1b) BASELINE.json configs[1] names an NPROMA sweep 32-256: the same NL bench at the other three blockings, each in a fresh
        # process with its own placed state (the headline's own blocking is this run's figure)
        sweep = {str(args.nproma): {"kernel_ms_avg": k_avg, "frac": fr["frac"], "value": value, "steps": args.steps, "source": "this run (the headline)"}}
        for npr in (32, 64, 128, 256):
            if npr == args.nproma:
                continue
            try:
                d = child_bench(["--kernel", "nl", "--steps", 100, "--warmup", 5, "--ngptot", args.ngptot, "--no-cpu-baseline", "--no-companions",
                                 "--nproma", npr, "--precision", args.precision] + (["--levapls2"] if args.levapls2 else []))
                sweep[str(npr)] = {"kernel_ms_avg": d["roofline"]["kernel_ms_avg"], "frac": d["roofline"]["frac"], "value": d["value"],
                                   "steps": d["steps"], "source": "child process"}
            except Exception as e:  # noqa: BLE001
                sweep[str(npr)] = {"error": repr(e)}
        out["nproma_sweep"] = dict(sorted(sweep.items(), key=lambda kv: int(kv[0])))
        # (2) north_star's target: >= 70 % of the HBM peak on the NL kernel at NGPTOT >= 1 M columns on one GPU
        try:
            d = child_bench(["--kernel", "nl", "--steps", 50, "--warmup", 5, "--ngptot", 1048576] + common)
            out["target_config"] = {"workload": d["config"]["workload"], "ngptot": 1048576, "steps": d["steps"], "value": d["value"],
                                    "unit": d["unit"], "ms_per_step": d["ms_per_step"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                                    "frac": d["roofline"]["frac"], "frac_kernel_only": d["roofline"].get("frac_kernel_only"),
                                    "frac_actual_bytes": d["roofline"].get("frac_actual_bytes"), "achieved": d["roofline"]["achieved"],
                                    "bytes_per_column": d["roofline"]["bytes_per_column"],
                                    "bytes_per_column_kernel_only": d["roofline"].get("bytes_per_column_kernel_only"),
                                    "placement": d["config"]["placement"],
                                    "target": "north_star: NL >= 0.70 of the 8 TB/s HBM3E peak at NGPTOT >= 1 M columns"}
        except Exception as e:  # noqa: BLE001
            out["target_config"] = {"error": repr(e)}
        # (3) what a caller gets WITHOUT the library's placement: first hipMalloc of a fresh process (CLOUDSC2_PLACE=0)
        try:
            d = child_bench(["--kernel", "nl", "--steps", 50, "--warmup", 5, "--ngptot", args.ngptot] + common, env={"CLOUDSC2_PLACE": "0"})
            out["roofline"]["unplaced_first_allocation"] = {"kernel_ms_avg": d["roofline"]["kernel_ms_avg"], "frac": d["roofline"]["frac"],
                                                            "note": "fresh process, CLOUDSC2_PLACE=0: plain first hipMalloc"}
        except Exception as e:  # noqa: BLE001
            out["roofline"]["unplaced_first_allocation"] = {"error": repr(e)}
        # (3b) what the UNCHANGED caller of CLOUDSC_DRIVER gets: host arrays through cloudsc2_nl_run, PCIe included
        try:
            out["host_array_driver"] = child_bench(["--host-driver-rate", "--ngptot", args.ngptot, "--nproma", args.nproma, "--precision", args.precision])
        except Exception as e:  # noqa: BLE001
            out["host_array_driver"] = {"error": repr(e)}
        # (4) the reference's two self-tests on a resident state of the same size: verdicts and the kernel time of the whole driver call
        try:
            out["self_tests"] = child_bench(["--self-tests", "--ngptot", args.ngptot, "--nproma", args.nproma, "--precision", args.precision])
        except Exception as e:  # noqa: BLE001
            out["self_tests"] = {"error": repr(e)}
        # (5) BASELINE.json configs[2] and configs[3] at their own sizes: the Taylor test at NGPTOT = 100 (the reference README's
        # `dwarf-cloudsc2-tl 1 100 1`) and the adjoint test at NGPTOT = 16 384, identity to 1e-12
        bc = {}
        try:
            d = child_bench(["--self-tests", "--ngptot", 100, "--nproma", 1, "--precision", args.precision])
            bc["configs[2] Taylor test, NGPTOT=100, NPROMA=1"] = d["taylor_test"]
            d = child_bench(["--self-tests", "--ngptot", 16384, "--nproma", args.nproma, "--precision", args.precision])
            a = d["adjoint_test"]
            a["identity_relative"] = a["znormg_in_eps"] * 2.220446049250313e-16
            a["to_1e-12"] = bool(a["identity_relative"] < 1e-12)
            bc[f"configs[3] adjoint test, NGPTOT=16384, NPROMA={args.nproma}"] = a
        except Exception as e:  # noqa: BLE001
            bc["error"] = repr(e)
        out["baseline_configs_2_3"] = bc
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.kernel == "nl":
        cb = cpu_baseline(tab, prm, 32, args.ngptot)
        if cb:
            out["cpu_baseline"] = cb
    if world > 1 and args.kernel == "nl" and not args.no_companions:
        # BASELINE.json configs[4] names NL + TL + AD on the node: every rank times the other two kernels IN-PROCESS on its own
        # columns (the NL state is given back first; 160 000 columns with their perturbation sets are 17 GB of the GPU's 288), with
        # the protocol of the headline -- barrier, K steps, barrier, MAX over ranks -- and `value` = all ranks' columns over that
        # time.  Same NPROMA as the headline for all three kernels (128: the TL / AD optimum of the one-GPU sweep, 1 % off NL's).
        # A failure here is recorded and does not touch the timing line that is already out.
        comp = {}
        try:
            del step, keep, ds
            torch.cuda.empty_cache()
            for kind in ("tl", "ad"):
                prm_k = c2.default_params(c2.ceta_from_table(tab), lregcl=(kind == "ad"), levapls2=args.levapls2)
                ds_k, step_k, bpc_k, keep_k, kname_k, place_k = build_step(c2, args, tab, prm_k, kind, dev, stream, col0)
                k_steps = max(5, min(args.steps, 30))
                el_k, kms_k = timed_steps(torch, dev, stream, step_k, k_steps, min(args.warmup, 5), barrier)
                el_k = float(c2dist.allreduce_max([el_k], dev)[0])
                per_rank = c2dist.allgather_scalar(float(kms_k.mean()), dev)
                worst = max(per_rank)
                comp[kind] = {"value": args.ngptot * world / (el_k / k_steps), "unit": "columns/s", "steps": k_steps, "ms_per_step": el_k / k_steps * 1e3,
                              "kernel_ms_avg": worst, "kernel_ms_avg_per_rank": [round(x, 5) for x in per_rank],
                              "bytes_per_column": bpc_k, "frac": bpc_k * args.ngptot / (worst * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_is": "the slowest rank's kernel against ONE GPU's HBM peak", "nproma": args.nproma, "kernel": kname_k,
                              "candidates_per_rank": [int(x) for x in c2dist.allgather_scalar(float(place_k.get("candidates", 0)), dev)]}
                if kind == "ad":
                    comp[kind].update(ad_design_floor(c2, ds_k.nlev, worst, args.ngptot, evap=args.levapls2))
                del step_k, keep_k, ds_k
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            comp["error"] = repr(e)
        out["companion_kernels"] = comp
        ds = step = keep = None
    native_hung = False
    if world > 1:
        # The only inter-GPU exchange of the path: max-reduce the two self-tests' verdict norms (outside the timing).  Each rank
        # runs the Taylor test and the adjoint test on its own 1024-column sub-range; ZNORMG(10) and ZNORMG are all-reduced
        # (MAX) over RCCL -- cloudsc_driver_tl_mod.F90:125, cloudsc_driver_ad_mod.F90:107 carried across ranks.
        try:
            ds = step = keep = None  # (gives the device memory back, whether or not the companions above already did)
            torch.cuda.empty_cache()
            ceta = c2.ceta_from_table(tab)
            vt = c2.state_from_table(tab, 64, 1024, col0=rank * 1024)
            ztl, _, _, _ = c2.run_state(c2.default_params(ceta, lregcl=False), vt, "tl")
            vt = c2.state_from_table(tab, 64, 1024, col0=rank * 1024)
            zad, _, _ = c2.run_state(c2.default_params(ceta, lregcl=True), vt, "ad")
            ztl_g = c2dist.allreduce_max(ztl, dev)
            zad_g = c2dist.allreduce_max([zad], dev)
            # the same two reductions once more through the native boundary the Fortran mains use (libcloudsc2_comm.so:
            # ncclAllReduce on a communicator bootstrapped by broadcasting the ncclUniqueId over torch.distributed)
            # -- in a thread with a deadline, so that a communicator that never comes up cannot take the JSON line with it
            native = {}

            def native_reductions():
                try:
                    from dwarf_p_cloudsc2_tl_ad_amd import comm as c2comm

                    torch.cuda.set_device(dev)
                    _, _, transport = c2comm.init_from_torch(local, dev)
                    n_tl = c2comm.allreduce(ztl, c2comm.MAX)
                    n_ad = c2comm.allreduce([zad], c2comm.MAX)
                    c2comm.finalize()
                    native.update({"transport": transport,
                                   "equals_torch_distributed": bool(np.array_equal(n_tl, ztl_g) and n_ad[0] == zad_g[0])})
                except Exception as e:  # noqa: BLE001
                    native.update({"error": repr(e)})

            import threading

            th = threading.Thread(target=native_reductions, daemon=True)
            th.start()
            native_wait = budgets(args.budget_s)[1]
            th.join(native_wait)
            if th.is_alive():
                native = {"error": f"no answer within {native_wait:g} s"}
                native_hung = True
            tl_ok, itest = c2.binding.taylor_verdict(ztl_g)
            out["verdicts"] = {"backend": torch.distributed.get_backend(),
                               "tl_znormg": [float(x) for x in ztl_g], "tl_passed": bool(tl_ok), "tl_penalty": int(itest),
                               "ad_symmetry_max_eps": float(zad_g[0]), "ad_ok": bool(c2.adjoint_verdict(float(zad_g[0]))),
                               "native_comm": native}
        except Exception as e:  # noqa: BLE001
            out["verdicts"] = {"error": repr(e)}
    if rank == 0:
        if world > 1 and not args.no_cpu_baseline and args.kernel == "nl" and not native_hung:
            # rank 0's host cores, after the last collective (no rank waits for it), only while the budget has room: the launcher's
            # deadline must not take the final line with it
            left = budgets(args.budget_s)[0] - (time.monotonic() - t_start)
            if left > 90.0:
                try:
                    cb = cpu_baseline(tab, prm, 32, args.ngptot, budget_s=min(15.0, left / 6.0))
                    if cb:
                        out["cpu_baseline"] = cb
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"] = {"error": repr(e)}
            else:
                out["cpu_baseline"] = {"skipped": f"{left:.0f} s of --budget-s left"}
        if world > 1:
            out["stage"] = "final"
            out["seconds_since_start"] = round(time.monotonic() - t_start, 1)
        print(json.dumps(out), flush=True)
    if native_hung:  # a thread of this rank still sits in the native communicator: leave without the orderly shutdown
        sys.stdout.flush()
        os._exit(3)  # the line is out, but a communicator that never answered is a failed run
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
