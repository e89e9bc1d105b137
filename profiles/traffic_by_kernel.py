"""Per-kernel HBM traffic of a whole benchmark run from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE cannot share a pass), summed per kernel symbol and grid.
Usage: python profiles/traffic_by_kernel.py <dir with pmc_*_fetch/ and pmc_*_write/ rocpd databases> <prefix>
FETCH_SIZE is in KiB and doubled (the gfx950 correction of the micro-architecture guide); WRITE_SIZE in KiB."""
import collections
import glob
import sqlite3
import sys

src, prefix = sys.argv[1], sys.argv[2]
tot = {}
for kind, cname, scale in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    db = sqlite3.connect(glob.glob(f"{src}/{prefix}_{kind}/*.db")[0])
    agg = collections.defaultdict(lambda: [0, 0.0])
    q = ("select k.name, k.grid_x, k.grid_y, k.grid_z, e.counter_name, e.counter_value from pmc_events e join kernels k "
         "on e.dispatch_id = k.dispatch_id")
    for name, gx, gy, gz, cn, val in db.execute(q):
        if cn == cname:
            a = agg[(name.split("(")[0].replace("void jirf::", ""), gx, gy, gz)]
            a[0] += 1
            a[1] += val * scale
    tot[kind] = agg
print("One launch shape per row: kernel symbol and grid (threads, as rocprofv3 reports it).  Reads = FETCH_SIZE x 2 "
      "(gfx950 correction), writes = WRITE_SIZE.\n")
print("| kernel | grid x,y,z | launches | read MB / launch | written MB / launch | total GB over the run |")
print("|---|---|---:|---:|---:|---:|")
rows = []
for k, (n, f) in tot["fetch"].items():
    w = tot["write"].get(k, (0, 0.0))[1]
    rows.append((f + w, k, n, f, w))
for t, k, n, f, w in sorted(rows, reverse=True):
    if t < 1e7:
        continue
    print(f"| `{k[0][:70]}` | {k[1]},{k[2]},{k[3]} | {n} | {f / n / 1e6:.1f} | {w / n / 1e6:.1f} | {t / 1e9:.2f} |")
