#!/usr/bin/env python3
"""Where a kernel's MFMAs, barriers, scratch accesses and waits sit in the compiler's assembly.
usage: isa_census.py file.s substring-of-mangled-name [first-line last-line]   (lines relative to the kernel's label)"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
start = next(k for k, l in enumerate(txt) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(k for k in range(start, len(txt)) if txt[k].startswith(".Lfunc_end"))
body = txt[start:end]
mf = [k for k, l in enumerate(body) if "v_mfma" in l]
sc = [k for k, l in enumerate(body) if "scratch_" in l]
bar = [k for k, l in enumerate(body) if "s_barrier" in l]
print(len(body), "lines; mfma", mf[0], "..", mf[-1], "(%d)" % len(mf), "; scratch ops", len(sc), "of them near the K loop:",
      [x for x in sc if mf[0] - 200 < x < mf[-1] + 50], "; barriers", bar)
if len(sys.argv) > 4:
    for k in range(int(sys.argv[3]), int(sys.argv[4])):
        print(k, body[k])
