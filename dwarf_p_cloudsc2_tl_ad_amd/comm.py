"""ctypes binding of include/cloudsc2_comm.h (libcloudsc2_comm.so): the dwarf's few collectives over RCCL -- the native
counterpart of the reference's cloudsc_mpi_mod (src/common/module/cloudsc_mpi_mod.F90) that the Fortran mains use.
The Python bench normally reduces through torch.distributed (dist.py); this module lets it exercise the native path too
(`init_from_torch`: the ncclUniqueId is broadcast by torch.distributed instead of through the rendezvous file)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLOUDSC2_COMM_LIB") or os.path.join(_HERE, "csrc", "libcloudsc2_comm.so")
SUM, MIN, MAX = 0, 1, 2
EXPORTED = ("cloudsc2_comm_init", "cloudsc2_comm_unique_id", "cloudsc2_comm_init_rank", "cloudsc2_comm_finalize",
            "cloudsc2_comm_rank", "cloudsc2_comm_size", "cloudsc2_comm_transport", "cloudsc2_comm_last_error",
            "cloudsc2_comm_allreduce_f64", "cloudsc2_comm_allreduce_i32", "cloudsc2_comm_allgather_i32", "cloudsc2_comm_barrier")
_lib = None


class CommError(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        try:  # one HIP runtime / one RCCL per process: let torch's copies be the ones that are mapped (see binding._load)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.cloudsc2_comm_transport.restype = C.c_char_p
        L.cloudsc2_comm_last_error.restype = C.c_char_p
        L.cloudsc2_comm_unique_id.argtypes = [C.c_char_p]
        L.cloudsc2_comm_init_rank.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.cloudsc2_comm_allreduce_f64.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int]
        L.cloudsc2_comm_allreduce_i32.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
        L.cloudsc2_comm_allgather_i32.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise CommError(f"cloudsc2_comm error {rc}: {(lib().cloudsc2_comm_last_error() or b'').decode()}")


def init():
    """Ranks from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK); returns (rank, world, transport)."""
    _check(lib().cloudsc2_comm_init())
    return lib().cloudsc2_comm_rank(), lib().cloudsc2_comm_size(), lib().cloudsc2_comm_transport().decode()


def init_from_torch(local_rank: int, device=None):
    """Inside an initialised torch.distributed job: rank 0 makes the ncclUniqueId, torch broadcasts its 128 bytes."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    buf = C.create_string_buffer(128)
    if rank == 0:
        _check(lib().cloudsc2_comm_unique_id(buf))
    dev = torch.device("cpu") if dist.get_backend() != "nccl" else (device or torch.device("cuda", torch.cuda.current_device()))
    t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    ident = bytes(t.cpu().tolist())
    _check(lib().cloudsc2_comm_init_rank(ident, rank, world, int(local_rank)))
    return rank, world, lib().cloudsc2_comm_transport().decode()


def finalize():
    _check(lib().cloudsc2_comm_finalize())


def allreduce(values, op: int = MAX) -> np.ndarray:
    v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)).copy())
    _check(lib().cloudsc2_comm_allreduce_f64(v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size), int(op)))
    return v


def allgather_i32(values) -> np.ndarray:
    v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.int32)))
    out = np.zeros(v.size * lib().cloudsc2_comm_size(), dtype=np.int32)
    _check(lib().cloudsc2_comm_allgather_i32(v.ctypes.data_as(C.POINTER(C.c_int)), int(v.size), out.ctypes.data_as(C.POINTER(C.c_int))))
    return out.reshape(lib().cloudsc2_comm_size(), v.size)
