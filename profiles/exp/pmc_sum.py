#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (rocpd SQLite):
   python profiles/exp/pmc_sum.py <dir with *_results.db below it> [kernel-name substring ...]
One row per (kernel, counter): dispatches, and the counter summed over its hardware instances, averaged over dispatches."""
import glob, os, sqlite3, sys
from collections import defaultdict

root, subs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
    db = sqlite3.connect(path)
    q = """select s.kernel_name, m.name, d.dispatch_id, e.value from rocpd_pmc_event e
           join rocpd_info_pmc m on e.pmc_id = m.id join rocpd_kernel_dispatch d on d.event_id = e.event_id
           join rocpd_info_kernel_symbol s on s.id = d.kernel_id"""
    for name, ctr, did, val in db.execute(q):
        if subs and not any(t in name for t in subs):
            continue
        short = name.split("(")[0]
        acc[short][ctr] += val
        disp[short][ctr].add((path, did))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = len(disp[k][c])
        print(f"    {c:28s} {acc[k][c] / n:18.0f}   per dispatch, {n} dispatches")
