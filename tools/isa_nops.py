#!/usr/bin/env python3
"""Which instruction pairs the compiler separated with s_nop in one kernel of a hipcc -S listing.
Usage: python3 tools/isa_nops.py build/isa/ntt.s fwd_rows8ILb1 [samples]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
ins = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
ctx = collections.Counter()
for i, l in enumerate(ins):
    if l.startswith("s_nop"):
        ctx[(ins[i - 1].split()[0], l, ins[i + 1].split()[0])] += 1
for k, v in ctx.most_common(16):
    print(v, k)
shown = set()
for i, l in enumerate(ins):
    if l.startswith("s_nop") and len(shown) < int(sys.argv[3]) if len(sys.argv) > 3 else False:
        k = (ins[i - 1].split()[0], ins[i + 1].split()[0])
        if k in shown:
            continue
        shown.add(k)
        print("-----", k)
        print("\n".join(ins[i - 5:i + 3]))
