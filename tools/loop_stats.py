#!/usr/bin/env python3
"""Instruction mix of the largest loop of one kernel: python tools/loop_stats.py KERNEL_MANGLED_SUBSTR [asm file]"""
import collections
import re
import sys

pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "dwarf_p_cloudsc2_tl_ad_amd/csrc/cloudsc2_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins, labels, br = [], {}, []
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        labels[m.group(1)] = len(ins)
    elif s and not s.startswith((";", ".")) and not s.endswith(":"):
        ins.append(s)
        if s.startswith(("s_cbranch", "s_branch")):
            br.append((len(ins), s.split()[-1]))
loops = sorted(((pos - labels[t], labels[t], pos) for pos, t in br if t in labels and labels[t] < pos), reverse=True)
print("static instructions", len(ins), "largest loops", loops[:3])
for length, a, b in loops[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    loop = ins[a:b]
    c = collections.Counter(i.split()[0] for i in loop)
    print(f"--- loop [{a},{b}) {length} instructions")
    for k, v in c.most_common(28):
        print(f"  {k:26s}{v}")
    print("  waits:", dict(collections.Counter(i for i in loop if i.startswith("s_waitcnt"))))
