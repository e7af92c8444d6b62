#!/usr/bin/env python3
"""Basic-block skeleton of one kernel: python tools/loop_cfg.py KERNEL_SUBSTR [asm]  -> per block: #instr, #valu, loads, stores, terminator"""
import re, sys
pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "dwarf_p_cloudsc2_tl_ad_amd/csrc/cloudsc2_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = []; cur = ["entry", []]
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    elif s and not s.startswith((";", ".")) and not s.endswith(":"):
        cur[1].append(s)
blocks.append(cur)
pos = 0
for name, ins in blocks:
    valu = sum(1 for i in ins if i.startswith("v_"))
    ld = sum(1 for i in ins if i.startswith("global_load")); st = sum(1 for i in ins if i.startswith("global_store"))
    sl = sum(1 for i in ins if i.startswith("s_load")); dv = sum(1 for i in ins if i.startswith("v_div_fixup"))
    br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print(f"{pos:5d} {name:12s} n={len(ins):4d} valu={valu:4d} ld={ld:2d} st={st:2d} sload={sl:2d} div={dv:2d}  {' | '.join(br)}")
    pos += len(ins)
