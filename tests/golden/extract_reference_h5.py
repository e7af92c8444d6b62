"""Extract the datasets the reference validates (cloudsc2_array_state_mod.F90:246-256) plus PLUDE from
config-files/reference.h5 into tests/golden/reference_h5.npz, using the HDF5 C API through ctypes (no h5py here).
Run in the build container only (needs /root/reference and /opt/conda/lib/libhdf5.so):

    python tests/golden/extract_reference_h5.py
"""
import ctypes as C
import os
import sys

import numpy as np

SRC = "/root/reference/config-files/reference.h5"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_h5.npz")
NAMES = ["PLUDE", "PCOVPTOT", "PFPLSL", "PFPLSN", "PFHPSL", "PFHPSN", "TENDENCY_LOC_A", "TENDENCY_LOC_Q", "TENDENCY_LOC_T",
         "TENDENCY_LOC_CLD"]


def main():
    h5 = C.CDLL("/opt/conda/lib/libhdf5.so")
    hid = C.c_int64
    h5.H5open()
    h5.H5Fopen.restype = hid
    h5.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    h5.H5Dopen2.restype = hid
    h5.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    h5.H5Dget_space.restype = hid
    h5.H5Dget_space.argtypes = [hid]
    h5.H5Sget_simple_extent_ndims.argtypes = [hid]
    h5.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    h5.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    native_double = hid.in_dll(h5, "H5T_NATIVE_DOUBLE_g").value
    f = h5.H5Fopen(SRC.encode(), 0, 0)
    if f < 0:
        sys.exit("cannot open " + SRC)
    out = {}
    for n in NAMES:
        d = h5.H5Dopen2(f, n.encode(), 0)
        sp = h5.H5Dget_space(d)
        nd = h5.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * nd)()
        h5.H5Sget_simple_extent_dims(sp, dims, None)
        a = np.zeros(tuple(int(x) for x in dims), dtype=np.float64)
        assert h5.H5Dread(d, native_double, 0, 0, 0, a.ctypes.data) >= 0
        out[n] = a
        print(n, a.shape, "nonzeros", int(np.count_nonzero(a)), "max|.|", float(np.abs(a).max()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
