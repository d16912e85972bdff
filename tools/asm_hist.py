#!/usr/bin/env python3
"""Instruction histogram of one kernel of a gfx950 .s file, split at s_barrier / branch labels.
usage: tools/asm_hist.py file.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    seg, segs, label = collections.Counter(), [], "entry"
    ops = collections.Counter()
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.LBB\S+):", s)
        if not m and (not s or s.startswith((";", "//", "."))):
            continue
        if m:
            if blocks:
                segs.append((label, seg))
                seg, label = collections.Counter(), m.group(1)
            continue
        op = s.split()[0]
        c = classify(op)
        seg[c] += 1
        ops[op] += 1
        if c == "barrier" and not blocks:
            segs.append((label, seg))
            seg, label = collections.Counter(), "after barrier"
    segs.append((label, seg))
    tot = collections.Counter()
    for name, c in segs:
        tot.update(c)
        if sum(c.values()) >= 8:
            print(f"{name:28s}", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("TOTAL", dict(tot))
    print("top ops:", ops.most_common(40))


main()
