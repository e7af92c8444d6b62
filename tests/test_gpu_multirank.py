"""The native multi-rank path (Fortran mains + fortran/cloudsc_mpi_mod.F90 + libcloudsc2_comm.so) and the device-resident
mode of the mains, on the one GPU of a test box:
  * two ranks (tools/launch_ranks.sh, CLOUDSC2_COMM=shm because RCCL refuses two ranks on one device) split NGPTOTG like
    dwarf_cloudsc.F90:64-69, run their sub-ranges, reduce the validation statistics (validate_mod.F90:197-199) and the
    verdict norms -- the reduced numbers must equal the one-rank run's;
  * the RCCL transport itself with a one-rank communicator (CLOUDSC2_COMM=rccl);
  * CLOUDSC2_RESIDENT=1: the state never exists on the host, results identical to the host-array drivers."""
from __future__ import annotations

import os
import re
import subprocess

import numpy as np
import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu
BLD = os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran", "build")
LAUNCH = os.path.join(ROOT, "tools", "launch_ranks.sh")


def _run(cmd, env=None, cwd=None, timeout=600):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, timeout=timeout, cwd=cwd,
                       env=None if env is None else {**os.environ, **env})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout, r.stderr


def _summary(out: str) -> dict:
    rows = {}
    for m in re.finditer(r"^\s*(PCOVPTOT|PFPLSL|PFPLSN|PFHPSL|PFHPSN|PA|TENDENCY_[TQLI])\s+(\S+)\s+(\S+)\s+(\S+)\s*$", out, flags=re.M):
        rows[m.group(1)] = tuple(float(x) for x in m.groups()[1:])
    assert len(rows) == 10, out
    return rows


def test_two_ranks_split_the_columns_and_reduce_the_statistics(tmp_path):
    exe = os.path.join(BLD, "dwarf-cloudsc2-nl")
    one, err1 = _run([exe, 1, 3000, 64], env={"CLOUDSC2_RESIDENT": "0"}, cwd=tmp_path)  # (the reference flow alone: one header, one table)
    two, err2 = _run([LAUNCH, 2, exe, 1, 3000, 64], env={"CLOUDSC2_COMM": "shm", "CLOUDSC2_RESIDENT": "0"}, cwd=tmp_path)
    assert "NUMPROC=1," in err1 and "NUMPROC=2," in err2
    assert "NGPBLKS=24" in err2                      # 1500 columns per rank (dwarf_cloudsc.F90:64-69) in blocks of 64
    assert err2.count("NUMPROC=2,") == 1             # rank 0 alone prints the header, and one table row per rank
    # the reference's timing table (timer_mod.F90:124-171): one worker row and one TOTAL row per rank, one grand TOTAL
    assert "@ rank#0:core#" in err2 and "@ rank#1:core#" in err2 and ": TOTAL @ rank#0" in err2 and ": TOTAL @ rank#1" in err2
    assert "      2 x 1" in err2 and "Time(msec)" in err2
    a, b = _summary(one), _summary(two)
    for k in a:  # min and max exactly; the sequential sums of 4e5 terms are added in another order
        assert a[k][0] == b[k][0] and a[k][1] == b[k][1], k
        assert abs(a[k][2] - b[k][2]) <= 1e-10 * abs(a[k][2]), k
    assert two.count("Variable") == 1


def test_two_ranks_reduce_the_verdict_norms(tmp_path):
    out1, _ = _run([os.path.join(BLD, "dwarf-cloudsc2-ad"), 1, 200, 100], cwd=tmp_path)
    out2, _ = _run([LAUNCH, 2, os.path.join(BLD, "dwarf-cloudsc2-ad"), 1, 200, 100], env={"CLOUDSC2_COMM": "shm"}, cwd=tmp_path)
    z1 = float(re.search(r"maximum error is\s+([0-9.Ee+-]+)", out1).group(1))
    z2 = float(re.search(r"maximum error is\s+([0-9.Ee+-]+)", out2).group(1))
    assert "TEST OK" in out1 and out2.count("TEST OK") == 1   # printed by rank 0 only
    assert z1 == z2                                           # the two ranks' columns are the one rank's: same maximum
    out2, _ = _run([LAUNCH, 2, os.path.join(BLD, "dwarf-cloudsc2-tl"), 1, 256, 32], env={"CLOUDSC2_COMM": "shm"}, cwd=tmp_path)
    assert out2.count("TEST PASSED, penalty") == 1 and out2.count("TL Taylor test") == 1


def test_rccl_transport_with_a_one_rank_communicator(tmp_path):
    """ncclCommInitRank / ncclAllReduce / ncclAllGather through libcloudsc2_comm.so on the box's one GPU."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dwarf_p_cloudsc2_tl_ad_amd import comm\n"
            "print(comm.init()); print(comm.allreduce([1.5, -2.0], comm.MAX).tolist(), comm.allreduce([3.0], comm.SUM).tolist(), "
            "comm.allgather_i32([7, 8]).tolist()); comm.finalize(); print('done')\n" % ROOT)
    out, _ = _run(["python3", "-c", code], env={"CLOUDSC2_COMM": "rccl", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, cwd=tmp_path)
    assert "(0, 1, 'rccl')" in out and "[1.5, -2.0] [3.0] [[7, 8]]" in out and "done" in out
    # and under the Fortran main: CLOUDSC_MPI_INIT -> RCCL communicator of one rank, reductions are identities
    out, err = _run([os.path.join(BLD, "dwarf-cloudsc2-ad"), 1, 100, 100], env={"CLOUDSC2_COMM": "rccl", "WORLD_SIZE": "1", "RANK": "0"},
                    cwd=tmp_path)
    assert "TEST OK" in out


def test_resident_mode_of_the_mains(tmp_path):
    exe = os.path.join(BLD, "dwarf-cloudsc2-nl")
    host, _ = _run([exe, 4, 16000, 32], env={"CLOUDSC2_RESIDENT": "0"}, cwd=tmp_path)
    res, err = _run([exe, 4, 16000, 32], env={"CLOUDSC2_RESIDENT": "1"}, cwd=tmp_path)
    assert "state resident on the GPU" in err
    a, b = _summary(host), _summary(res)
    for k in a:
        assert a[k] == b[k], (k, a[k], b[k])
    # the resident rate is the kernel's, not PCIe's: the mean of ten back-to-back launches
    m = re.search(r"GPU kernel\s+([0-9.]+) ms =\s+([0-9.Ee+-]+) columns/s", err)
    assert m and float(m.group(2)) > 2e7, err
    out, _ = _run([os.path.join(BLD, "dwarf-cloudsc2-tl"), 1, 100, 1], env={"CLOUDSC2_RESIDENT": "1"}, cwd=tmp_path)
    assert "TEST PASSED, penalty" in out
    out, _ = _run([os.path.join(BLD, "dwarf-cloudsc2-ad"), 1, 100, 100], env={"CLOUDSC2_RESIDENT": "1"}, cwd=tmp_path)
    assert "TEST OK" in out
    # two resident ranks
    out, err = _run([LAUNCH, 2, exe, 1, 3000, 64], env={"CLOUDSC2_COMM": "shm", "CLOUDSC2_RESIDENT": "1"}, cwd=tmp_path)
    one, _ = _run([exe, 1, 3000, 64], cwd=tmp_path)
    a, b = _summary(one), _summary(out)
    for k in a:
        assert a[k][0] == b[k][0] and a[k][1] == b[k][1] and abs(a[k][2] - b[k][2]) <= 1e-10 * abs(a[k][2]), k
    # the default mode (the sweep resident, then the reference flow, both rates printed) with two ranks: one summary, rank 0's two lines
    out, err = _run([LAUNCH, 2, exe, 1, 3000, 64], env={"CLOUDSC2_COMM": "shm"}, cwd=tmp_path)
    b = _summary(out)
    for k in a:
        assert a[k][0] == b[k][0] and a[k][1] == b[k][1] and abs(a[k][2] - b[k][2]) <= 1e-10 * abs(a[k][2]), k
    assert err.count("NUMPROC=2,") == 2 and err.count("state resident on the GPU (cloudsc2_state_*") == 1, err
    assert err.count("CLOUDSC_DRIVER on host arrays (the reference flow, PCIe-bound)") == 1, err


def test_resident_validation_against_a_reference_file(tmp_path):
    """CLOUDSC2_RESIDENT=1 with a reference.h5 in the working directory: VALIDATE runs on the device (cloudsc2_state_validate)
    and prints the reference's table; the file is written by a host-array run of the same configuration, so every error is 0."""
    exe = os.path.join(BLD, "dwarf-cloudsc2-nl")
    _run([exe, 1, 1000, 100], env={"CLOUDSC2_WRITE_REFERENCE": "1"}, cwd=tmp_path)
    assert os.path.exists(tmp_path / "reference.h5")
    host, _ = _run([exe, 1, 1000, 100], cwd=tmp_path)
    res, _ = _run([exe, 1, 1000, 100], env={"CLOUDSC2_RESIDENT": "1"}, cwd=tmp_path)
    rows_h = [ln for ln in host.splitlines() if re.search(r"\d D\d|\dD\d", ln)]
    rows_r = [ln for ln in res.splitlines() if re.search(r"\d D\d|\dD\d", ln)]
    assert len(rows_h) == 10 and rows_h == rows_r, (rows_h, rows_r)
    assert not any("!!!!" in ln for ln in rows_r)
