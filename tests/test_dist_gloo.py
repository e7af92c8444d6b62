"""N>1 host logic on CPU: two processes over gloo (the GPU path uses the same calls over RCCL, backend "nccl").
Columns shard as contiguous sub-ranges (the reference's MPI split, dwarf_cloudsc.F90:66-69) with no data-path
collective; the only exchange is the max-reduction of the two self-tests' verdict norms and the NL validation sums."""
from __future__ import annotations

import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, world: int, port: int, ngptotg: int, nproma: int, out_dir: str):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist

    import dwarf_p_cloudsc2_tl_ad_amd as c2
    from dwarf_p_cloudsc2_tl_ad_amd import dist as c2dist

    r, _, w = c2dist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    col0, n = c2dist.shard(ngptotg, rank, world)
    tab = c2.synthetic_table()
    st = c2.state_from_table(tab, nproma, n, col0=col0)
    # this rank's columns, in global order
    cols = st.PT.transpose(0, 2, 1).reshape(-1, st.nlev)[:n]
    np.save(os.path.join(out_dir, f"pt_{rank}.npy"), cols)
    # verdict reductions: element-wise max over ranks (cloudsc_driver_tl_mod.F90:125 / cloudsc_driver_ad_mod.F90:107)
    local_tl = np.array([1.0 + 10.0 ** (-(k + 1)) * (rank + 1) for k in range(10)])
    got_tl = c2dist.allreduce_max(local_tl)
    got_ad = c2dist.allreduce_max([7.5 + rank])
    stats = {"PFPLSN": {"min": -1.0 - rank, "max": 2.0 + rank, "maxabserr": 1e-15 * (rank + 1), "sumabserr": 1e-14,
                        "sumabsref": 1.0 + rank}}
    red = c2dist.allreduce_validation(stats)
    np.save(os.path.join(out_dir, f"red_{rank}.npy"),
            np.concatenate([got_tl, got_ad, [red["PFPLSN"][k] for k in ("min", "max", "maxabserr", "sumabserr", "sumabsref")]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_verdict_reduction(tmp_path):
    import torch.multiprocessing as mp

    import dwarf_p_cloudsc2_tl_ad_amd as c2

    world, ngptotg, nproma = 2, 333, 32
    port = _free_port()
    mp.spawn(_worker, args=(world, port, ngptotg, nproma, str(tmp_path)), nprocs=world, join=True)

    # the union of the shards is the global state: no column lost, duplicated or re-ordered
    tab = c2.synthetic_table()
    glob = c2.state_from_table(tab, nproma, ngptotg)
    want = glob.PT.transpose(0, 2, 1).reshape(-1, glob.nlev)[:ngptotg]
    got = np.concatenate([np.load(tmp_path / f"pt_{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(want, got)
    assert c2.column_range(ngptotg, 0, 2) == (0, 167) and c2.column_range(ngptotg, 1, 2) == (167, 333)

    r0, r1 = np.load(tmp_path / "red_0.npy"), np.load(tmp_path / "red_1.npy")
    assert np.array_equal(r0, r1)                                   # every rank holds the reduced verdicts
    assert np.allclose(r0[:10], [1.0 + 10.0 ** (-(k + 1)) * 2 for k in range(10)])
    assert r0[10] == 8.5
    assert list(r0[11:]) == [-2.0, 3.0, 2e-15, 2e-14, 3.0]          # min / max / max / sum / sum (validate_mod.F90:197-199)


def test_single_process_reductions_are_identity():
    from dwarf_p_cloudsc2_tl_ad_amd import dist as c2dist

    assert np.array_equal(c2dist.allreduce_max([3.0, 1.0]), [3.0, 1.0])
    assert c2dist.shard(160000, 3, 8) == (60000, 20000)
