"""libcloudsc2_comm.so (include/cloudsc2_comm.h), the native counterpart of the reference's cloudsc_mpi_mod: every declared
symbol is exported; one rank is a no-op; three ranks over the shared-memory rehearsal transport reduce and gather exactly
like CLOUDSC_MPI_REDUCE_{SUM,MIN,MAX} / CLOUDSC_MPI_GATHER (cloudsc_mpi_mod.F90:102-327).  The RCCL transport itself needs
one GPU per rank: it is exercised with one rank in tests/test_gpu_comm.py and by bench.py --gpus N on a multi-GPU node."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

from tests.util import ROOT

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from dwarf_p_cloudsc2_tl_ad_amd import comm
rank, world, transport = comm.init()
r = float(rank)
out = {"rank": rank, "world": world, "transport": transport,
       "max": comm.allreduce([1.0 + r, -r, 10.0 * r], comm.MAX).tolist(),
       "min": comm.allreduce([1.0 + r, -r], comm.MIN).tolist(),
       "sum": comm.allreduce([0.1 * (r + 1)] * 10, comm.SUM).tolist(),
       "gather": comm.allgather_i32([rank, 100 + rank]).tolist()}
for _ in range(50):  # many small collectives back to back: the barrier's sense reversal
    v = comm.allreduce([r], comm.SUM)
out["loop"] = v.tolist()
comm.finalize()
print("RESULT " + json.dumps(out), flush=True)
""" % ROOT


def test_comm_library_exports_every_declared_symbol():
    from dwarf_p_cloudsc2_tl_ad_amd import comm

    header = open(os.path.join(ROOT, "include", "cloudsc2_comm.h")).read()
    declared = set(re.findall(r"\b(cloudsc2_comm_\w+)\s*\(", header))
    assert declared == set(comm.EXPORTED), declared ^ set(comm.EXPORTED)
    lib = C.CDLL(comm.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_single_rank_is_a_no_op():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json

    d = json.loads(r.stdout.split("RESULT ", 1)[1])
    assert d["transport"] == "single" and d["world"] == 1 and d["max"] == [1.0, -0.0, 0.0] and d["gather"] == [[0, 100]]


def test_three_ranks_reduce_and_gather_over_the_rehearsal_transport(tmp_path):
    import json

    world = 3
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29888",
                   CLOUDSC2_COMM="shm", CLOUDSC2_COMM_TOKEN=f"pytest_{os.getpid()}")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    res = sorted((json.loads(so.split("RESULT ", 1)[1]) for so, _ in outs), key=lambda d: d["rank"])
    for d in res:
        assert d["transport"] == "shm" and d["world"] == 3
        assert d["max"] == [3.0, 0.0, 20.0] and d["min"] == [1.0, -2.0]
        assert np.allclose(d["sum"], [0.6] * 10, rtol=1e-15)
        assert d["gather"] == [[0, 100], [1, 101], [2, 102]]
        assert d["loop"] == [3.0]
    assert res[0]["sum"] == res[1]["sum"] == res[2]["sum"]  # fixed reduction order: identical bits on every rank


def test_a_leftover_segment_of_a_crashed_job_is_not_mistaken_for_this_one():
    """Same rendezvous name as a job that died (no MASTER_PORT under mpirun, a reused launcher pid): the stale POSIX segment is
    there before any rank starts, full of another launch's counters.  Ranks != 0 must not attach to it (they used to, and then
    waited for ranks that never come): the segment carries the launch's nonce, rank 0 replaces it, the others wait for theirs."""
    import json

    world = 2
    # no CLOUDSC2_COMM_TOKEN: the name comes from MASTER_PORT + this process's pid, the nonce also from this process's start time
    name = f"/_tmp_cloudsc2_comm_29889_{os.getpid()}.shm"
    path = "/dev/shm" + name
    with open(path, "wb") as f:  # what a crashed 4-rank job left: wrong nonce, everybody "attached", barrier mid-flight
        f.write((0x1234567812345678).to_bytes(8, "little") + (3).to_bytes(4, "little") + (1).to_bytes(4, "little") + (4).to_bytes(4, "little")
                + (4).to_bytes(4, "little") + bytes(64 * 64 * 8))
    try:
        procs = []
        for rank in (1, 0):  # rank 1 first: it finds the stale segment before rank 0 has replaced it
            env = {k: v for k, v in os.environ.items() if k not in ("CLOUDSC2_COMM_TOKEN", "TORCHELASTIC_RUN_ID")}
            env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29889", CLOUDSC2_COMM="shm")
            procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
            if rank == 1:
                import time

                time.sleep(1.5)
        outs = [p.communicate(timeout=180) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, se[-2000:]
        for so, _ in outs:
            d = json.loads(so.split("RESULT ", 1)[1])
            assert d["world"] == 2 and d["max"] == [2.0, 0.0, 10.0] and d["loop"] == [1.0]
    finally:
        if os.path.exists(path):
            os.unlink(path)
