#!/usr/bin/env python3
"""Static VALU count and issue-cycle estimate of one kernel, split into regions of a source file given as
'label=substring of the line where the region starts' (compile with -gline-tables-only -S).  Cycle weights: the
measured issue costs of tools/harness/ubench_valu.hip on MI355X (2.5 cycles for f32 fma/mul/add and u32 add, 8.3
for transcendentals, 4.3 for everything else incl. all fp64, DPP, conversions, packed ops; MFMA 16.5).
usage: tools/asm_regions.py file.s <kernel regex> <source file> label=marker ..."""
import collections
import re
import sys


def weight(op):
    if op.startswith("v_mfma"):
        return 16.5
    if op.startswith(("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_add_u32", "v_sub_u32", "v_fmac_f32", "v_subrev_u32")):
        return 2.5
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos", "v_exp", "v_log")):
        return 8.3
    return 4.3


def main():
    path, kre, srcfile = sys.argv[1:4]
    src = open(srcfile).read().split("\n")
    marks = []
    for a in sys.argv[4:]:
        label, sub = a.split("=", 1)
        marks.append((label, next(i + 1 for i, l in enumerate(src) if sub in l)))
    marks.sort(key=lambda t: t[1])
    base = srcfile.split("/")[-1]
    lines = open(path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and re.search(kre, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    cur = (0, 0)
    cnt, cyc, lds, vm = (collections.Counter() for _ in range(4))
    for l in lines[start:end]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        fn = files.get(cur[0], "?")
        if fn == base:
            reg = "(before first marker)"
            for label, st in marks:
                if cur[1] >= st:
                    reg = label
        else:
            reg = "other: " + fn
        if op.startswith("v_"):
            cnt[reg] += 1
            cyc[reg] += weight(op)
        elif op.startswith("ds_"):
            lds[reg] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            vm[reg] += 1
    tot = sum(cyc.values()) or 1
    for k in sorted(set(cnt) | set(lds) | set(vm)):
        print(f"{k:44s} valu {cnt[k]:5d}  cycles {cyc[k]:7.0f} ({100 * cyc[k] / tot:4.1f} %)  lds {lds[k]:4d}  vmem {vm[k]:3d}")
    print(f"total valu {sum(cnt.values())}  estimated issue cycles {tot:.0f}")


main()
