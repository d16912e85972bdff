"""Per-basic-block instruction mix of a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import re, sys
lbl, rows, cur = "entry", [], None
def flush():
    if cur and cur["n"]: rows.append((lbl, cur))
cur = dict(n=0, pk=0, ds=0, vm=0, s=0, v=0, f64=0, line=0)
for ln, line in enumerate(open(sys.argv[1]), 1):
    m = re.match(r"^(\.LBB[0-9_]+):", line)
    if m:
        flush(); lbl = m.group(1); cur = dict(n=0, pk=0, ds=0, vm=0, s=0, v=0, f64=0, line=ln); continue
    m = re.match(r"^\s+((v|s|ds|global|buffer|flat|scratch)_\w+)", line)
    if not m: continue
    op = m.group(1); cur["n"] += 1
    if re.match(r"v_pk_(fma|mul)_f32", op): cur["pk"] += 1
    if op.startswith("ds_"): cur["ds"] += 1
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vm"] += 1
    if op.startswith("s_"): cur["s"] += 1
    if op.startswith("v_"): cur["v"] += 1
    if "f64" in op: cur["f64"] += 1
flush()
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print("label line n valu pkfma f64 lds vmem salu")
for l, c in rows:
    if c["n"] >= thr: print(l, c["line"], c["n"], c["v"], c["pk"], c["f64"], c["ds"], c["vm"], c["s"])
