#!/usr/bin/env python3
"""Instruction histogram of every loop (backward branch) of one kernel in a hipcc -S listing, with the wait states
the compiler had to fill with s_nop.  Usage: python3 tools/isa_loop.py build/isa/keyswitch.s ks_rows_mac"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l]
    for start in starts:
        end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].strip() == "s_endpgm")
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        print("==", lines[start].rstrip(":"))
        spans = loops or [(0, len(body) - 1)]
        for a, b in spans:
            c = collections.Counter()
            waits = 0
            for l in body[a:b + 1]:
                l = l.strip()
                if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
                    continue
                op = l.split()[0]
                c[op] += 1
                if op == "s_nop":
                    waits += int(l.split()[1]) + 1
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print(f"   {'loop' if loops else 'body'} lines {a}-{b}: {sum(c.values())} instructions, VALU {valu}, s_nop {c['s_nop']} ({waits} wait states)")
            if len(sys.argv) > 3:
                for k, v in c.most_common(int(sys.argv[3])):
                    print(f"      {k} {v}")


if __name__ == "__main__":
    main()
