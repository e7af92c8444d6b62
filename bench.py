#!/usr/bin/env python
"""bench.py -- CLOUDSC2 NL hot path (SATUR + CLOUDSC2 over all NPROMA blocks) on N MI355X GPUs.

One "step" = one pass of the NL kernel over this rank's NGPTOT=160000 columns x 137 levels (fp64), inputs resident
in HBM.  Weak scaling: every rank owns its own 160000-column sub-range of the global columns (the reference's MPI
split); there is no collective in the data path.  Prints ONE JSON line (rank 0).

Before the measurement the state is placed: which physical HBM an allocation lands in decides up to 17 % of the kernel
time on this part, so candidates are allocated in several regions of the 288 GB (--placement-regions),
each is timed with ten launches, and the fastest is used for the W warm-up and K timed steps.  Every candidate's
time is in config.placement; --placement-regions 0 takes whatever the first allocation gets.

The line also carries `cpu_baseline` (the reference on the host cores, rank 0 at N=1) and, at N=1, `companion_kernels`:
the same bench for TL and AD, each run as a child process after the NL measurement (BASELINE.json's metric names all
three kernels); neither is inside the timed region.
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


def spawn_ranks(ngpus: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) of this same script and
    relay rank 0's JSON line.  Runs before anything in this process has touched the GPU or imported torch; the children get
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exactly as `python -m torch.distributed.run --nproc-per-node N` would set
    them, so both ways of starting the bench run the same code."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), LOCAL_WORLD_SIZE=str(ngpus))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    procs = []
    for r in range(ngpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    if lines:
        print(lines[-1], flush=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or not lines:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ngptot", type=int, default=160000, help="columns per GPU")
    ap.add_argument("--nproma", type=int, default=128)
    ap.add_argument("--kernel", choices=["nl", "tl", "ad"], default="nl")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true", help="skip the short TL and AD timings appended to the NL line")
    ap.add_argument("--precision", choices=["double", "single"], default=os.environ.get("CLOUDSC2_PRECISION", "double"),
                    help="single = the fp32 library (the reference's -DSINGLE build); the headline metric is double")
    ap.add_argument("--levapls2", action="store_true", help="switch the evaporation branch on (off in every shipped config)")
    ap.add_argument("--placement-regions", default="0,9,18,27,36,45,54,63,72,81,90,99,108,117,126,135,144,153,162,171,180,189,198,207",
                    help="GiB offsets in HBM at which candidate placements of the state are timed before the measurement "
                         "(the fastest is used; '0' = just allocate)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # no launcher: become one (nothing here has touched the GPU yet)
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")

    import torch

    os.environ["CLOUDSC2_PRECISION"] = args.precision  # read by the package at import: one precision per process
    import dwarf_p_cloudsc2_tl_ad_amd as c2
    from dwarf_p_cloudsc2_tl_ad_amd import dist as c2dist

    single = c2.binding.SINGLE
    fp = "fp32" if single else "fp64"
    variant = ", the -DSINGLE variant" if single else ""

    rank, local, world = c2dist.init_process_group()
    if not torch.cuda.is_available() or not c2.device_available():
        raise SystemExit("bench.py needs a HIP device: the CLOUDSC2 engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab), lregcl=(args.kernel == "ad"), levapls2=args.levapls2)
    col0 = rank * args.ngptot  # weak scaling: rank r owns global columns [r*NGPTOT, (r+1)*NGPTOT)
    stream = torch.cuda.current_stream(dev)

    def make_workload():
        """The state (tiled on the device from the 100-column table, cloudsc2_expand_launch: no host copy exists) and
        whatever else the kernel under test touches; returns (step, bytes per column, kernel name, keep-alive)."""
        ds = c2.DeviceState.from_table(tab, args.nproma, args.ngptot, dev, start=col0)
        nlev = ds.nlev
        if args.kernel == "nl":
            return (lambda: ds.nl(prm, stream)), c2.bytes_per_column(nlev, "nl_driver"), \
                "nl_kernel<F> (SATUR + CLOUDSC2 fused; fast math, no evaporation branch, 32-bit offsets when buffers < 4 GiB)", ds
        ds.satur(prm, stream)
        inc = ds.increments(zero_supsat=(args.kernel == "ad"))
        dout = c2.FlatFields("out", ds.nb, ds.nlev, ds.nproma, dev)
        if args.kernel == "tl":
            return (lambda: ds.tl(prm, inc, dout, stream)), c2.bytes_per_column(nlev, "tl"), \
                "tl_kernel<C2F_QSAT> (CLOUDSC2TL, trajectory recomputed, not stored)", (ds, inc, dout)
        ds.tl(prm, inc, dout, stream)
        scratch = ds.new_scratch()
        # + carry checkpoint plane (write + read)
        return (lambda: ds.ad(prm, inc, dout, scratch, stream)), c2.bytes_per_column(nlev, "ad") + 2 * c2.binding.REAL_BYTES * nlev, \
            "ad_kernel<C2F_QSAT> (CLOUDSC2AD: trajectory pass + reverse pass)", (ds, inc, dout, scratch)

    # Where in the 288 GB of HBM the state lives decides up to 17 % of the kernel time (stable for the life of an allocation,
    # different per box: DESIGN.md 5, tools/placement_probe.py).  As a long-running model would at start-up, candidates are
    # placed in several regions of the memory (spacer allocations in between, never touched), each is timed, the fastest is
    # used; everything stays allocated so the chosen placement is not disturbed.  All candidates' times are reported.
    regions = [float(x) for x in args.placement_regions.split(",") if x.strip() != ""] or [0.0]
    cands, trial_ms, spacers = [], [], []
    for r_gib in regions:
        used = torch.cuda.memory_reserved(dev) / 2**30
        if r_gib > used + 1.0:
            try:
                spacers.append(torch.empty(int((r_gib - used) * 2**30), dtype=torch.uint8, device=dev))
            except RuntimeError:
                break  # not that much memory left: stop exploring
        try:
            w = make_workload()
        except RuntimeError:  # out of device memory: keep the candidates there are
            torch.cuda.empty_cache()
            if not cands:
                raise
            break
        cands.append(w)
    for w in cands:  # timed once all candidates exist: a fresh allocation needs ~10 launches to reach its steady time
        for _ in range(5):
            w[0]()
        torch.cuda.synchronize(dev)
        tev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in tev:
            a.record(stream)
            w[0]()
            b.record(stream)
        torch.cuda.synchronize(dev)
        trial_ms.append(sorted(a.elapsed_time(b) for a, b in tev)[2])  # median of 5 launches
    best = min(range(len(cands)), key=lambda i: trial_ms[i])
    step, bpc, kname, keep = cands[best]
    nlev = (keep[0] if isinstance(keep, tuple) else keep).nlev

    def barrier():
        if torch.distributed.get_backend() == "nccl":
            torch.distributed.barrier(device_ids=[local])  # the rank's own GPU, not a guess
        else:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = float(c2dist.allreduce_max([elapsed], dev)[0])  # MAX over ranks (RCCL all-reduce of one double)
    kms = np.array([a.elapsed_time(b) for a, b in ev])  # per-launch device time on the launch stream
    ms_per_step = elapsed / args.steps * 1e3
    total_cols = args.ngptot * world
    value = total_cols / (elapsed / args.steps)

    k_avg = float(kms.mean())
    achieved = bpc * args.ngptot / (k_avg * 1e-3) / 1e9
    # HBM traffic from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, calibrated as the MI355X guide
    # prescribes; tools/pmc_workload.py + tools/pmc_parse.py), scaled per column to this launch.  null if not measured.
    traffic = None
    pmc_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_traffic.json")) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    pmc_files = [f for f in pmc_files if ("_sp_" in f) == single]  # counters are per precision
    if pmc_files:
        try:
            pmcs = [json.load(open(os.path.join(ROOT, "profiles", f))) for f in pmc_files]
            same = [p for p in pmcs if p.get("ngptot") == args.ngptot]  # prefer the pass taken at this launch size
            pmc = (same or pmcs)[-1]
            traffic = pmc["kernels"][args.kernel]["traffic_bytes"] / pmc["ngptot"] * args.ngptot
        except (KeyError, ValueError, OSError):
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_column": bpc,
                "algorithmic_bytes": bpc * args.ngptot, "kernel_ms_avg": k_avg, "kernel_ms_min": float(kms.min())}

    out = {
        "metric": f"CLOUDSC2 {args.kernel.upper()} columns/sec ({fp}, NLEV=137)", "value": value, "unit": "columns/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if single else "f64", "data": "synthetic",
        "config": {"workload": f"CLOUDSC2 {args.kernel.upper()} {fp}, NGPTOT={args.ngptot} columns per GPU, NLEV=137, "
                               f"NPROMA={args.nproma} (BASELINE.json configs[1]{variant})",
                   "ngptot_per_gpu": args.ngptot, "nlev": nlev, "nproma": args.nproma,
                   "parallelism": f"columns sharded over {world} GPU(s), no data-path collective",
                   "device": device_info(torch, dev),
                   "placement": {"regions_gib": regions[:len(trial_ms)], "kernel_ms_per_candidate": [round(x, 4) for x in trial_ms],
                                 "chosen": best}},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and args.kernel == "nl" and not args.no_companions:
        # BASELINE.json's metric names NL/TL/AD: the same bench for the other two kernels, each in a child process with its
        # own placement search, after this process has given its device memory back.  Outside the timed region, never
        # part of `value`; `--no-companions` skips it.
        import subprocess

        del cands, step, keep, spacers, w
        torch.cuda.empty_cache()
        comp = {}
        for kind in ("tl", "ad"):
            cmd = [sys.executable, os.path.abspath(__file__), "--kernel", kind, "--steps", "30", "--warmup", "5", "--no-cpu-baseline",
                   "--ngptot", str(args.ngptot), "--nproma", str(args.nproma), "--precision", args.precision,
                   "--placement-regions", args.placement_regions] + (["--levapls2"] if args.levapls2 else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                comp[kind] = {"value": d["value"], "unit": d["unit"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                              "bytes_per_column": d["roofline"]["bytes_per_column"], "frac": d["roofline"]["frac"],
                              "traffic": d["roofline"]["traffic"], "kernel": d["roofline"]["kernel"],
                              "placement_ms": d["config"]["placement"]["kernel_ms_per_candidate"]}
            except Exception as e:  # noqa: BLE001  (never let the companions break the headline line)
                comp[kind] = {"error": repr(e)}
        out["companion_kernels"] = comp
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.kernel == "nl":
        cb = cpu_baseline(tab, prm, 32, args.ngptot)
        if cb:
            out["cpu_baseline"] = cb
    if world > 1:
        # the only inter-GPU exchange of the path: max-reduce the self-test verdict norms over RCCL (outside the timing)
        try:
            vt = c2.state_from_table(tab, 64, 1024, col0=rank * 1024)
            zad, _, _ = c2.run_state(c2.default_params(c2.ceta_from_table(tab), lregcl=True), vt, "ad")
            znorm = c2dist.allreduce_max([zad], dev)
            out["verdicts"] = {"ad_symmetry_max_eps": float(znorm[0]), "ad_ok": bool(c2.adjoint_verdict(float(znorm[0])))}
        except Exception as e:  # noqa: BLE001
            out["verdicts"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
