#!/usr/bin/env python3
"""The reference's own NL acceptance table (CLOUDSC2_ARRAY_STATE_VALIDATE) for the GPU kernels against the reference
Fortran's outputs on the 100 synthetic columns (tests/golden/nl_synth100.npz stands in for reference.h5, which cannot be
reproduced without input.h5):  python tools/validate_report.py [NGPTOT] [NPROMA]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import dwarf_p_cloudsc2_tl_ad_amd as c2  # noqa: E402


def golden_reference_table(tab):
    g = np.load(os.path.join(ROOT, "tests", "golden", "nl_synth100.npz"))
    z = np.zeros_like(g["out_tent"])
    return {"PLUDE": tab["PLUDE"], "PCOVPTOT": g["out_covptot"], "PFPLSL": g["out_fplsl"], "PFPLSN": g["out_fplsn"],
            "PFHPSL": g["out_fhpsl"], "PFHPSN": g["out_fhpsn"], "TENDENCY_LOC_A": z, "TENDENCY_LOC_Q": g["out_tenq"],
            "TENDENCY_LOC_T": g["out_tent"], "TENDENCY_LOC_CLD": np.stack([g["out_tenl"], g["out_teni"], z, z, z])}


def main():
    ngptot = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    nproma = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    tab = c2.synthetic_table()
    prm = c2.default_params(c2.ceta_from_table(tab))
    ref = golden_reference_table(tab)
    for precise in (False, True):
        c2.set_math_mode(precise)
        ds = c2.DeviceState.from_table(tab, nproma, ngptot)
        ds.nl(prm)
        _, text = ds.validate(ref)
        print(f"--- math mode: {'precise' if precise else 'fast'}; NGPTOT={ngptot} NPROMA={nproma}")
        print(text)
    c2.set_math_mode(False)


if __name__ == "__main__":
    main()
