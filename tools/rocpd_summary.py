#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 (ROCm 7.x, rocpd sqlite) result database.
usage: tools/rocpd_summary.py <results.db>   ->   name, calls, total_us, avg_us, pct"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    print(f"\"{name}\",{calls},{total:.3f},{avg:.3f},{pct:.2f}")
try:
    rows = list(c.execute("select name, count(*), avg(vgpr_count), avg(sgpr_count), avg(lds_size), avg(scratch_size), "
                          "avg(grid_x), avg(grid_y), avg(workgroup_x) from kernels group by name"))
    print("\nkernel,dispatches,vgpr,sgpr,lds,scratch,grid_x,grid_y,wg_x")
    for r in rows:
        print(",".join(f"\"{v}\"" if isinstance(v, str) else f"{v:g}" for v in r))
except sqlite3.Error as e:
    print("# (no per-dispatch register info:", e, ")")
