"""Per (kernel symbol, grid) device time of the AST optimisation step from a rocprofv3 kernel trace.
Usage: python profiles/train_by_grid.py <rocpd .db> [steps]   (prints a table, longest first)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = {}
for name, gx, gy, gz, wx, dur in db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration from kernels"):
    a = agg.setdefault((name, gx // max(wx, 1), gy, gz), [0, 0])
    a[0] += 1
    a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot / 1e6 / steps:.3f} ms per step over {steps} steps")
for (name, gx, gy, gz), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    short = name.replace("jirf::(anonymous namespace)::", "").replace("jirf::", "")
    short = short[:short.index("(")] if "(" in short and not short.startswith("void") else short[:70]
    print(f"{t / 1e6 / steps:8.3f} ms/step {n // steps:4d} x {t / n / 1e3:8.1f} us  grid {gx}x{gy}x{gz}  {short[:80]}")
