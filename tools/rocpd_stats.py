"""Per-kernel statistics from a rocprofv3 (rocpd sqlite) --kernel-trace result: calls, average / min / max duration.
   python tools/rocpd_stats.py <results.db> [csv_out]
   ROCPD_BY_GRID=<substring>: kernels whose name contains it are additionally split by launch grid (one projection shape each)."""
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
q = f"select s.{name_col}, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 6 desc"
rows = list(cur.execute(q))
tot = sum(r[5] for r in rows)
lines = ["kernel,calls,avg_us,min_us,max_us,total_ms,pct"]
for n, c, a, mn, mx, t in rows:
    lines.append(f"\"{n}\",{c},{a/1e3:.3f},{mn/1e3:.3f},{mx/1e3:.3f},{t/1e6:.3f},{100*t/tot:.2f}")
sub = os.environ.get("ROCPD_BY_GRID")
if sub:
    gcol = [c for c in cols if c in ("grid_size_x", "grid_x", "workgroup_count_x")]
    if gcol:
        q2 = f"select s.{name_col}, d.{gcol[0]}, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id where s.{name_col} like '%{sub}%' group by s.{name_col}, d.{gcol[0]} order by 1, 2"
        for n, g, c, a, mn, mx in cur.execute(q2):
            lines.append(f"\"{n} grid={g}\",{c},{a/1e3:.3f},{mn/1e3:.3f},{mx/1e3:.3f},,")
    else:
        lines.append("# no grid column among: " + " ".join(cols))
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
