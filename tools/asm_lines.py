#!/usr/bin/env python3
"""Per-source-line instruction counts of one kernel (compile with -gline-tables-only -S).
usage: tools/asm_lines.py file.s <kernel name substring> [file substring]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else ""
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
cur = (0, 0)
cnt = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end + 1]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith((";", "//", ".")) or s.endswith(":"):
        continue
    op = s.split()[0]
    k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else \
        "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "s"
    cnt[cur] += 1
    kinds[cur][k] += 1
tot = collections.Counter()
for (f, ln), c in sorted(cnt.items()):
    fn = files.get(f, "?")
    if want in fn:
        print(f"{fn.split('/')[-1]}:{ln:5d} {c:5d}  " + " ".join(f"{k}={v}" for k, v in sorted(kinds[(f, ln)].items())))
    tot.update(kinds[(f, ln)])
print(dict(tot))
