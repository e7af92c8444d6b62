"""Real-data readiness, pinned mechanically.

`config-files/input.h5` is not distributed with the reference checkout (`.MISSING_LARGE_BLOBS`), so the NL comparison with
`config-files/reference.h5` cannot run yet.  What CAN be pinned today is that the day the file appears nothing has to be
edited: the loaders of this build (csrc/cloudsc2_io.cpp `kScalars`, fileio.py, fortran/cloudsc2_hip_state_mod.F90) ask the file
for exactly the dataset names -- and shapes -- the reference's own loader asks for, restricted to what the hot path reads.

The expectation is EXTRACTED from the reference's sources as text (study, not execution; nothing is copied into the repo):
  * every `LOAD_SCALAR('<dataset>', <variable>)` of yomcst.F90 / yoethf.F90 / yoecldp.F90 / yoephli.F90,
  * of those, the ones whose variable is READ by the hot path: the executable statements of satur / cuadjtqs / cloudsc2 /
    cuadjtqstl / cloudsc2tl / cuadjtqsad / cloudsc2ad plus the statement functions of fcttre*.func.h they reach (transitively),
  * every `LOAD_AND_EXPAND('<dataset>', ...)` of CLOUDSC2_ARRAY_STATE_LOAD (cloudsc2_array_state_mod.F90:164-199) and the four
    dataset suffixes of LOAD_AND_EXPAND_STATE (expand_mod.F90:151-154), with the level count each is read with.
CPU-only; skipped where /root/reference does not exist (the GPU box)."""
from __future__ import annotations

import os
import re

import pytest

from tests.util import ROOT

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on this box")

KERNELS = (("cloudsc2_nl", ("satur", "cuadjtqs", "cloudsc2")), ("cloudsc2_tl", ("cuadjtqstl", "cloudsc2tl", "cloudsc2")),
           ("cloudsc2_ad", ("cuadjtqsad", "cloudsc2ad")))


def _code(text: str) -> str:
    """Executable text: comments, preprocessor lines and USE statements dropped, continuation lines joined, upper case."""
    lines = []
    for ln in text.splitlines():
        if ln.lstrip().startswith(("!", "#")):
            continue
        lines.append(ln.split("!")[0])
    joined = re.sub(r"&\s*\n\s*&?", " ", "\n".join(lines)).upper()
    return "\n".join(ln for ln in joined.splitlines() if not re.match(r"\s*USE\b", ln))


def _loaded_scalars() -> dict:
    """dataset name -> Fortran variable, for the four *_LOAD_PARAMETERS routines."""
    out = {}
    for mod in ("yomcst", "yoethf", "yoecldp", "yoephli"):
        for ln in open(f"{REF}/common/module/{mod}.F90"):
            m = re.search(r"LOAD_SCALAR\('(\w+)',\s*([\w%]+)\)", ln)
            if m:
                out[m.group(1)] = m.group(2).upper()
    return out


def _hot_path_text() -> str:
    body = "\n".join(_code(open(f"{REF}/{d}/{f}.F90").read()) for d, fs in KERNELS for f in fs)
    funcs = {}  # statement functions of the include files: name -> right-hand side
    for h in ("fcttre", "fcttretl", "fcttread", "fccld"):
        for ln in _code(open(f"{REF}/common/include/{h}.func.h").read()).splitlines():
            m = re.match(r"\s*(\w+)\s*\(([^)]*)\)\s*=\s*(.*)$", ln)
            if m and not re.match(r"\s*REAL", ln):
                funcs[m.group(1)] = m.group(3)
    used, todo = set(), [f for f in funcs if re.search(rf"\b{f}\s*\(", body)]
    while todo:
        f = todo.pop()
        if f in used:
            continue
        used.add(f)
        todo += [g for g in funcs if g not in used and re.search(rf"\b{g}\s*\(", funcs[f])]
    assert {"FOEALFA", "FOEEWM"} <= used  # the saturation formulas (fcttre.func.h:74,...)
    return body + "\n" + "\n".join(funcs[f] for f in used)


def test_the_scalar_datasets_asked_for_are_exactly_what_the_hot_path_reads():
    loaded = _loaded_scalars()
    assert len(loaded) > 150 and loaded["RG"] == "RG" and loaded["YRECLDP_RCLCRIT"] == "YRECLDP%RCLCRIT"
    text = _hot_path_text()
    need = {n for n, v in loaded.items() if re.search(r"(?<![\w%])" + re.escape(v) + r"\b", text)}
    # YREPHLI%LPHYLIN is loaded and then overwritten by every main before the first kernel call (dwarf_cloudsc.F90:106-107 of
    # cloudsc2_nl / _tl / _ad: "overload LPHYLIN"): a flag of cloudsc2_params (lphylin, default 1), not a dataset this build reads
    for d in ("cloudsc2_nl", "cloudsc2_tl", "cloudsc2_ad"):
        assert re.search(r"^\s*YREPHLI%LPHYLIN\s*=\s*\.true\.", open(f"{REF}/{d}/dwarf_cloudsc.F90").read(), flags=re.M | re.I)
    need.discard("YREPHLI_LPHYLIN")
    io = open(os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "csrc", "cloudsc2_io.cpp")).read()
    asked = re.findall(r'\{"(\w+)",\s*&cloudsc2_params::(\w+)\}', io)
    names = [a for a, _ in asked]
    assert len(names) == len(set(names))
    assert set(names) == need, (sorted(set(names) - need), sorted(need - set(names)))
    # and each lands in the parameter of the same name (RCLCRIT <- YRECLDP_RCLCRIT, RLPTRC <- YREPHLI_RLPTRC, RG <- RG ...)
    for ds, field in asked:
        assert ds.split("_", 1)[1].lower() == field if ds.startswith(("YRECLDP_", "YREPHLI_")) else ds.lower() == field, (ds, field)
    # the scalars of the state loader itself (cloudsc2_array_state_mod.F90:162-163,193)
    for ds in ("KLON", "KLEV", "PTSPHY"):
        assert f'"{ds}"' in io, ds


def _reference_fields() -> dict:
    """dataset name -> ('KLEV' | 'KLEV+1', ndim) as CLOUDSC2_ARRAY_STATE_LOAD reads them."""
    src = open(f"{REF}/common/module/cloudsc2_array_state_mod.F90").read()
    out = {}
    for m in re.finditer(r"CALL LOAD_AND_EXPAND\('(\w+)',\s*SELF%\w+,\s*KLON,\s*SELF%(KLEV(?:\+1)?),\s*(NCLV,)?", src):
        out[m.group(1)] = (m.group(2), 5 if m.group(3) else 1)
    m = re.search(r"CALL LOAD_AND_EXPAND_STATE\('(\w+)'", src)
    ex = open(f"{REF}/common/module/expand_mod.F90").read()
    for sfx, nd in re.findall(r"load_array\(name//'(_\w+)',\s*start,\s*end,\s*size,\s*nlon,\s*nlev,\s*(ndim,)?", ex):
        out[m.group(1) + sfx] = ("KLEV", 5 if nd else 1)
    return out


def test_the_field_datasets_asked_for_are_the_reference_loaders():
    want = _reference_fields()
    assert set(want) == {"PT", "PQ", "PAP", "PAPH", "PLU", "PLUDE", "PMFU", "PMFD", "PA", "PCLV", "PSUPSAT", "TENDENCY_CML_T",
                         "TENDENCY_CML_A", "TENDENCY_CML_Q", "TENDENCY_CML_CLD"}, sorted(want)
    assert want["PAPH"] == ("KLEV+1", 1) and want["PCLV"] == ("KLEV", 5) and want["TENDENCY_CML_CLD"] == ("KLEV", 5)
    # the Python loader
    from dwarf_p_cloudsc2_tl_ad_amd import fileio

    py = open(fileio.__file__).read()
    asked_py = set(fileio.INPUT_FIELDS_2D) | set(re.findall(r'f\.read\("(\w+)"\)', py[py.index("def read_input_file"):py.index("def reference_table_from_state")]))
    assert asked_py == set(want), (sorted(asked_py ^ set(want)))
    # the Fortran loader of the mains (both the host-array and the device-resident path ask for every dataset)
    f90 = open(os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran", "cloudsc2_hip_state_mod.F90")).read()
    tables = re.findall(r"READ_TABLE\(F,\s*'(\w+)',\s*(TABH?)\)", f90)
    bulk = re.findall(r"CLOUDSC2_FILE_READ_REAL\(F,\s*'(\w+)'//C_NULL_CHAR,\s*TAB3", f90)
    for path in (0, 1):  # the two LOAD variants list the datasets in the same order: first half, second half
        half = len(tables) // 2
        asked = {n for n, _ in tables[path * half:(path + 1) * half]} | set(bulk)
        assert asked == set(want), (path, sorted(asked ^ set(want)))
    for n, buf in tables:  # half-level datasets go through the (KLON, KLEV+1) buffer
        assert (buf == "TABH") == (want[n][0] == "KLEV+1"), n
    for n in bulk:
        assert want[n][1] == 5, n


def test_the_reference_file_datasets_compared_are_the_validators():
    """VALIDATE (cloudsc2_array_state_mod.F90:246-256) compares these ten datasets of reference.h5; WRITE_REFERENCE (:275-284)
    writes them.  The checkout's reference.h5 holds them with the shapes this build reads (tests/test_fileio.py reads the file)."""
    src = open(f"{REF}/common/module/cloudsc2_array_state_mod.F90").read()
    written = set(re.findall(r"CALL WRITE_ARRAY\('(\w+)'", src))
    from dwarf_p_cloudsc2_tl_ad_amd import fileio

    assert written == set(fileio.REFERENCE_FIELDS), sorted(written ^ set(fileio.REFERENCE_FIELDS))
    f90 = open(os.path.join(ROOT, "dwarf_p_cloudsc2_tl_ad_amd", "fortran", "cloudsc2_hip_state_mod.F90")).read()
    checked = set(re.findall(r"CALL CHECK2\('(\w+)'", f90)) | set(re.findall(r"CLOUDSC2_FILE_READ_REAL\(F,\s*'(TENDENCY_LOC_\w+)'", f90))
    assert checked == written, sorted(checked ^ written)
