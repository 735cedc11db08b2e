#!/usr/bin/env python3
"""Static instruction mix of every loop of a gfx950 .s file (hipcc -S --cuda-device-only): what a loop body costs in
VALU / SALU / LDS / memory / spill instructions.  Development aid for the kernels' register budgets."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
lab = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = i
loops = {}
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\S*\s+(?:\S+,\s*)?(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        s = lab[m.group(1)]
        loops[s] = max(loops.get(s, 0), i)


def mix(s, e):
    c = collections.Counter()
    for l in lines[s:e + 1]:
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        if op.startswith(("v_readlane", "v_writelane")):
            k = "lane_rw"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_waitcnt"):
            k = "wait"
        elif op.startswith("s_barrier"):
            k = "barrier"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_bpermute"):
            k = "bperm"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith("scratch"):
            k = "scratch"
        elif op.startswith(("global", "buffer", "flat")):
            k = "vmem"
        else:
            k = "other"
        c[k] += 1
    return c


minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for s, e in sorted(loops.items()):
    if e - s < minlen:
        continue
    c = mix(s, e)
    print(f"loop @{s + 1}-{e + 1} ({e - s} lines): " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
