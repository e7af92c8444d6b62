#!/usr/bin/env python3
"""Static instruction mix of one kernel in the hipcc -S output (csrc/cloudsc2_kernels.s): python tools_isa_count.py nl_kernelILb0ELb0E"""
import collections
import re
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "nl_kernelILb0ELb0E"
path = sys.argv[2] if len(sys.argv) > 2 else "dwarf_p_cloudsc2_tl_ad_amd/csrc/cloudsc2_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = [l.strip().split()[0] for l in lines[start + 1:end]
       if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
c = collections.Counter(ins)
print("total static instructions", len(ins))
groups = collections.Counter()
for k, v in c.items():
    if k.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")): groups["fp64 arith"] += v
    elif k.startswith(("v_rcp_f64", "v_sqrt_f64", "v_rsq_f64")): groups["fp64 transcendental"] += v
    elif k.startswith(("v_div_", )): groups["fp64 div helpers"] += v
    elif k.startswith(("v_readlane", "v_writelane")): groups["sgpr spill traffic"] += v
    elif k.startswith(("global_load", "global_store", "scratch_")): groups["vmem"] += v
    elif k.startswith("s_load"): groups["smem"] += v
    elif k.startswith("s_"): groups["salu/ctrl"] += v
    elif k.startswith(("v_cndmask", "v_cmp")): groups["cmp/select"] += v
    elif k.startswith(("v_mov", "v_accvgpr")): groups["moves"] += v
    else: groups["other valu"] += v
for k, v in groups.most_common(): print(f"  {k:24s}{v}")
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25): print(f"{k:28s}{v}")
